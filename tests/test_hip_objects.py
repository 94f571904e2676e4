"""-m gpu: ragged object lists with device-resident counts (csrc/objects_pad.hip), the fused bin-edge kernel, a captured hipGraph
that takes LIVE objects and the validation step as one 2B-image forward (item 1a) -- against the reference's
torch formulation (modules/ObjCAViT.py:180-194), the CPU oracle, and eager dispatch."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import gen
from oracle import restate
from objcavit_amd.config import make_args
from util import max_rel, rel_dev

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
PAD = 0.0001


@pytest.fixture(scope="module")
def ops():
    from objcavit_amd import hip_ops
    return hip_ops


@pytest.mark.parametrize("counts,cap,S,group,nmax", [([5, 3, 1], 5, 20, None, 0), ([5, 3, 1], 8, 20, None, 0), ([2, 7, 4, 1], 7, 12, 2, 0),
                                                    ([3, 2], 6, 9, None, 6), ([9, 9], 9, 9, None, 0), ([1], 1, 132, None, 0),
                                                    ([70, 12, 5, 9], 70, 132, 2, 0)])
def test_object_pad_kernels_vs_reference_formulation(ops, counts, cap, S, group, nmax):
    """ocv_object_tokens_pad_fwd / ocv_object_front_pad_fwd against pad_sequence + F.pad as the reference writes them, per GROUP of
    images (one group = one call of the reference: its Nmax is that call's longest list) and with a caller-given Nmax."""
    B, E = len(counts), 128
    rows = [gen.randn(f"o{i}", (n, E), 5) for i, n in enumerate(counts)]
    tok = torch.full((B, cap, E), float("nan"))                          # rows beyond the count: garbage (NaN on purpose)
    for i, r in enumerate(rows):
        tok[i, :counts[i]] = r
    cnt = torch.tensor(counts, dtype=torch.int32).cuda()
    out, mask = ops.object_tokens_pad(tok.cuda(), cnt, PAD)
    ref = torch.nn.utils.rnn.pad_sequence(rows, batch_first=True, padding_value=PAD)
    ref = F.pad(ref, (0, 0, 0, cap - ref.shape[1]), value=PAD)
    ref_mask = torch.arange(cap)[None] >= torch.tensor(counts)[:, None]
    assert torch.equal(out.cpu(), ref) and torch.equal(mask.cpu().bool(), ref_mask)
    enc = gen.randn("enc", (B, cap, E), 6)                               # what the object encoder hands on (padded rows hold values too)
    keys, kpm = ops.object_front_pad(enc.cuda(), cnt, S, PAD, group=group, nmax=nmax)
    g = group or B
    for lo in range(0, B, g):                                            # the reference, one call per group
        c = counts[lo:lo + g]
        n = nmax if nmax else max(c)
        amt = S - n
        r_keys = F.pad(enc[lo:lo + g, :n], (0, 0, amt, 0), value=PAD)                                      # :194
        r_mask = F.pad(torch.arange(n)[None] >= torch.tensor(c)[:, None], (0, amt), value=True)            # :193
        assert torch.equal(keys[lo:lo + g].cpu(), r_keys) and torch.equal(kpm[lo:lo + g].cpu().bool(), r_mask)
    with pytest.raises(ValueError):
        ops.object_front_pad(enc.cuda(), cnt, cap - 1, PAD)             # more object rows than image tokens


def test_object_pad_stray_counts_keep_one_defined_key(ops):
    """device counts nobody validated.  A count of 0 (or below) keeps ONE live key so that no image's softmax is fully
    masked -- and that key is the PAD row, not whatever row 0 of the caller's buffer holds; a count beyond the capacity is the
    capacity.  (Host-side lists are validated before they reach the device: PaddedObjects.from_lists rejects empty lists.)"""
    B, cap, E, S = 4, 6, 128, 20
    tok = gen.randn("tok", (B, cap, E), 9)
    cnt = torch.tensor([0, -3, 99, 2], dtype=torch.int32).cuda()
    out, mask = ops.object_tokens_pad(tok.cuda(), cnt, PAD)
    out, mask = out.cpu(), mask.cpu().bool()
    for b in (0, 1):
        assert bool((out[b] == PAD).all()) and mask[b].tolist() == [False] + [True] * (cap - 1)
    assert torch.equal(out[2], tok[2]) and not bool(mask[2].any())
    assert torch.equal(out[3, :2], tok[3, :2]) and bool((out[3, 2:] == PAD).all()) and mask[3].tolist() == [False, False] + [True] * (cap - 2)
    keys, kpm = ops.object_front_pad(tok.cuda(), cnt, S, PAD)
    assert kpm.cpu().bool()[0].tolist() == [False] + [True] * (S - 1) and not bool(kpm.cpu().bool()[2, :cap].any())


@pytest.mark.parametrize("B,n,norm", [(1, 256, "linear"), (16, 256, "linear"), (3, 100, "sigmoid"), (2, 1000, "linear"), (2, 7, "none")])
def test_bin_edges_kernel_vs_torch(ops, B, n, norm):
    raw = gen.randn("raw", (B, n), 3, 2.0)
    if norm == "none":
        raw = torch.softmax(raw, 1)
    w, e, c = ops.bin_edges(raw.cuda(), norm, 0.001, 80.0)
    y = raw.double()
    y = torch.relu(y) + 0.1 if norm == "linear" else (torch.sigmoid(y) if norm == "sigmoid" else y)
    wr = y / y.sum(1, keepdim=True) if norm != "none" else y
    er = torch.cumsum(F.pad((80.0 - 0.001) * wr, (1, 0), value=0.001), 1)
    cr = 0.5 * (er[:, :-1] + er[:, 1:])
    assert rel_dev(w, wr.float()) < 1e-6 and rel_dev(e, er.float()) < 1e-6 and rel_dev(c, cr.float()) < 1e-6
    assert bool((e[:, 0] == 0.001).all())


def _model(kw, H, W, seed, provider=None):
    from objcavit_amd.modules.GraphBins import GraphBins
    args = make_args(dimensions_train=[H, W], dimensions_test=[H, W], language="clip", **kw)
    m = GraphBins(args, object_provider=provider).eval()
    sd = gen.load_into(m, seed, gen.PEAKY)
    return m.cuda(), sd, args


def _objects(counts, seed, H, W):
    feats = [gen.randn(f"f{seed}_{i}", (1 if n is None else n, 512), seed, 10.0 / np.sqrt(512)) for i, n in enumerate(counts)]
    xywh = [None if n is None else gen.boxes(f"b{seed}_{i}", n, seed, H, W) for i, n in enumerate(counts)]
    return feats, xywh


@pytest.mark.parametrize("kw,S_note", [(dict(strategy="learned"), "Q1 layout"), (dict(strategy="learned_bbox_wh", use_2_saca=True), "Q3"),
                                       (dict(strategy="grid_random_roi_align"), "configs[4]'s strategy")])
def test_one_captured_graph_serves_live_ragged_object_sets(kw, S_note):
    """ONE hipGraph captured for (B = 3, capacity 100) replayed with three different ragged object sets --
    incl. an image without detections (the <UNK> row with the box (-1, -1, -1, -1)), more objects than S / 2 (Q1: real rows reach
    the unmasked key positions) and the second SA/CA stack (Q3: Nmax-dependent) -- equals eager dispatch at the same capacity bit
    for bit, eager dispatch at the batch's own capacity to rounding, and the CPU oracle to the north-star bar."""
    from objcavit_amd.graph import GraphedGraphBins
    from objcavit_amd.modules.ObjCAViT import PaddedObjects
    H, W, B, cap = 352, 384, 3, 100                       # S = 11 x 12 = 132 tokens
    m, sd, _ = _model(kw, H, W, 17)
    img = gen.randn("img", (B, 3, H, W), 17).cuda()
    g = GraphedGraphBins(m, img, object_capacity=cap)
    cu = lambda ts: [None if t is None else t.cuda() for t in ts]          # noqa: E731
    # (under roi_align the reference's "no detections" box (-1, -1, -1, -1) has no extent: 0 / 0 = NaN for that image, there and
    # here -- DESIGN.md section 2 -- so that strategy gets a two-object image in its place)
    none = 2 if kw["strategy"] == "grid_random_roi_align" else None
    for s, counts in enumerate(([16, 5, 9], [none, 100, 3], [1, 1, 70])):
        feats, xywh = _objects(counts, 40 + s, H, W)
        im = gen.randn(f"img{s}", (B, 3, H, W), 18 + s).cuda()
        r = g(im, cu(feats), cu(xywh))
        rd, re = r.depth_pred.clone(), r.bin_edges.clone()
        po = PaddedObjects.from_lists(cu(feats), cu(xywh), im.device, capacity=cap)
        same = m(im, po)
        assert torch.equal(rd, same.depth_pred) and torch.equal(re, same.bin_edges), (kw, counts)
        own = m(im, cu(feats), cu(xywh))
        assert rel_dev(re, own.bin_edges) < 1e-5 and max_rel(rd, own.depth_pred) < 1e-4
        ref_d, ref_e = restate.graphbins_forward(im.cpu(), feats, xywh, sd, 0.001, 10, **kw)
        assert rel_dev(re, ref_e) < 1e-4 and max_rel(rd, ref_d) < 1e-3, (kw, counts)
    with pytest.raises(ValueError):
        g(img, cu(_objects([101, 1, 1], 50, H, W)[0]), cu(_objects([101, 1, 1], 50, H, W)[1]))      # beyond the captured capacity
    baked = GraphedGraphBins(m, img)
    with pytest.raises(RuntimeError):
        baked(img, cu(feats), cu(xywh))                                    # captured with its objects baked in


def test_table_object_provider_under_a_captured_graph():
    """The graph asks the model's provider for every replay's image (detections change per image; the provider runs OUTSIDE the
    graph, its output is copied into the static buffers): TableObjectProvider with ragged detections incl. an image without any."""
    from objcavit_amd.graph import GraphedGraphBins
    from objcavit_amd.objects import TableObjectProvider
    H, W, seed = 352, 384, 91
    table = gen.randn("table", (40, 512), seed, 10.0 / np.sqrt(512)).cuda()
    state = {"k": 0}
    dets = [([gen.boxes("b0", 5, seed, H, W), None, gen.boxes("b2", 1, seed, H, W)], [torch.tensor([3, 17, 17, 39, 0]), None, torch.tensor([8])]),
            ([None, gen.boxes("c1", 12, seed, H, W), gen.boxes("c2", 2, seed, H, W)], [None, torch.arange(12), torch.tensor([1, 2])])]

    def detector(image):
        xy, cl = dets[state["k"]]
        return [None if b is None else b.to(image.device) for b in xy], cl

    prov = TableObjectProvider(detector, class_table=table)
    m, sd, _ = _model(dict(strategy="learned"), H, W, seed, provider=prov)
    img = gen.randn("img", (3, 3, H, W), seed).cuda()
    g = GraphedGraphBins(m, img, object_capacity=16)
    for k in (1, 0, 1):
        state["k"] = k
        r = g(img)
        rd = r.depth_pred.clone()
        feats, boxes, _ = prov(img)
        ref_d, ref_e = restate.graphbins_forward(img.cpu(), [f.cpu() for f in feats], [None if b is None else b.cpu() for b in boxes],
                                                 sd, 0.001, 10, strategy="learned")
        assert rel_dev(r.bin_edges, ref_e) < 1e-4 and max_rel(rd, ref_d) < 1e-3, k


@pytest.mark.parametrize("kw", [dict(strategy="learned"), dict(strategy="learned_bbox_wh", use_2_saca=True)])
def test_validation_step_as_one_2b_forward(kw):
    """image + mirror as ONE 2B-image forward (object_group = B keeps each half's own Nmax) against the two
    forwards the reference issues (modules/GraphBinsLM.py:159,173): the records agree to rounding (batch-size-dependent kernel
    dispatch), also when the detector finds different object counts in the mirrored images and the second SA/CA stack is on."""
    from objcavit_amd.validation import ValidationStep
    H, W, B = 352, 384, 2
    pool = [16, 5, 9, 90]                                                  # object counts handed out to images in order of first sight

    class Prov:                                                            # a "detector": the same image always gets the same objects,
        def __init__(self):                                                # an image and its mirror get different ones
            self.seen = {}

        def __call__(self, image):
            ramp = torch.arange(image.shape[3], device=image.device, dtype=torch.float32)
            f, b = [], []
            for i in range(image.shape[0]):
                key = round(float((image[i, 0, 0] * ramp).sum()), 2)
                if key not in self.seen:
                    k = len(self.seen)
                    fs, bs = _objects([pool[k]], 60 + k, H, W)
                    self.seen[key] = (fs[0].to(image.device), bs[0].to(image.device))
                f.append(self.seen[key][0])
                b.append(self.seen[key][1])
            return f, b, None

    m, _, args = _model(kw, H, W, 23, provider=Prov())
    img = gen.randn("img", (B, 3, H, W), 23).cuda()
    gt = (torch.rand(B, 1, H, W, generator=torch.Generator().manual_seed(3)) * 9.0 + 0.5).cuda()
    rec_j, out_j = ValidationStep(m, args, joint=True)(img, gt, first_image_id=4)
    assert len(m.object_provider.seen) == 2 * B
    rec_t, out_t = ValidationStep(m, args, joint=False)(img, gt, first_image_id=4)
    assert tuple(out_j.depth_pred.shape) == tuple(out_t.depth_pred.shape) == (B, 1, H // 2, W // 2)
    assert max_rel(out_j.depth_pred, out_t.depth_pred) < 1e-4 and rel_dev(out_j.bin_edges, out_t.bin_edges) < 1e-5
    assert torch.equal(rec_j[:, 8:], rec_t[:, 8:]) and rel_dev(rec_j[:, :8], rec_t[:, :8]) < 1e-4


@pytest.mark.parametrize("joint_capture", [False, True])
def test_validation_step_on_a_captured_graph(joint_capture):
    """ValidationStep(GraphedGraphBins(...)) -- a graph captured for B images serves image and mirror as two replays, one
    captured for 2B images with object_group = B as the joint forward; both give the eager model's records."""
    from objcavit_amd.graph import GraphedGraphBins
    from objcavit_amd.modules.GraphBins import SyntheticObjectProvider
    from objcavit_amd.validation import ValidationStep
    H, W, B = 352, 384, 2
    m, _, args = _model(dict(strategy="learned_bbox_wh", use_2_saca=True), H, W, 37, provider=SyntheticObjectProvider(12, "clip", seed=5))
    img = gen.randn("img", (B, 3, H, W), 37).cuda()
    gt = (torch.rand(B, 1, H, W, generator=torch.Generator().manual_seed(5)) * 9.0 + 0.5).cuda()
    rec_e, out_e = ValidationStep(m, args, joint=joint_capture)(img, gt, first_image_id=2)
    d_e, e_e = out_e.depth_pred.clone(), out_e.bin_edges.clone()
    if joint_capture:
        g = GraphedGraphBins(m, torch.cat([img, img.flip(dims=[3])], 0), object_group=B)
    else:
        g = GraphedGraphBins(m, img)
    rec_g, out_g = ValidationStep(g, args, joint=True)(img, gt, first_image_id=2)
    assert tuple(out_g.depth_pred.shape) == (B, 1, H // 2, W // 2)
    assert max_rel(out_g.depth_pred, d_e) < 1e-5 and rel_dev(out_g.bin_edges, e_e) < 1e-6
    assert torch.equal(rec_g[:, 8:], rec_e[:, 8:]) and rel_dev(rec_g[:, :8], rec_e[:, :8]) < 1e-5


@pytest.mark.parametrize("B", [1, 2])
def test_baseline_config_at_the_reference_validation_batch(B):
    """BASELINE configs[2] (NYU 480x640, learned positional MLP, 32 objects with text features) at the batch the reference's own
    validation loop runs (main.py:58 forces bs 1; image + mirror = 2): every image against the CPU oracle, replay == eager."""
    from objcavit_amd.graph import GraphedGraphBins
    from objcavit_amd.modules.GraphBins import SyntheticObjectProvider
    H, W = 480, 640
    m, sd, _ = _model(dict(strategy="learned"), H, W, 31, provider=SyntheticObjectProvider(32, "clip", seed=9))
    img = gen.randn("img", (B, 3, H, W), 31).cuda()
    out = m(img)
    d, e = out.depth_pred.clone(), out.bin_edges.clone()
    feats, boxes, _ = m.object_provider(img)
    for i in range(B):
        ref, ref_e = restate.graphbins_forward(img[i:i + 1].cpu(), [feats[i].cpu()], [boxes[i].cpu()], sd, 0.001, 10.0, strategy="learned")
        assert rel_dev(e[i:i + 1], ref_e) < 1e-4 and max_rel(d[i:i + 1], ref) < 1e-3, i
    g = GraphedGraphBins(m, img)
    assert torch.equal(g(img).depth_pred, d)


def test_independent_streams_do_not_share_a_hardware_queue():
    """Round 6: slots in flight need streams on hardware queues of their own, and the runtime's own dealing does not give that (of the
    streams created in a row, the third and the fourth shared a queue: profiles/r06_stream_queues.txt).  hip_ops.independent_streams
    checks pairwise with a long empty kernel on one stream and a tiny one on the other."""
    from objcavit_amd import hip_ops
    dev = torch.device("cuda:0")
    sts = hip_ops.independent_streams(4, dev)
    assert len(sts) == 4 and len({s.cuda_stream for s in sts}) == 4
    assert hip_ops.ROUTE_REPORT.get("independent_streams") is None          # tests/conftest.py: GPU_MAX_HW_QUEUES = 4
    for i in range(4):
        for j in range(4):
            if i != j:
                assert not hip_ops.streams_share_a_queue(sts[i], sts[j]), (i, j)
    assert hip_ops.streams_share_a_queue(sts[0], sts[0])                    # (the probe sees what it should: one stream is one queue)


def test_pipelined_validation_equals_the_sequential_step():
    """PipelinedValidation: the reference's bs-1 validation loop with four steps in flight (the default: one captured joint image +
    mirror graph per slot, own stream each) gives the records of ValidationStep(joint=True) issued one after the other -- nine images,
    more than two rounds of the four slots, live objects from the model's provider (different per image and for the mirror)."""
    from objcavit_amd.validation import PipelinedValidation, ValidationStep
    H, W, B, N = 352, 384, 1, 9

    class Prov:                                             # a "detector": objects follow from the image content
        def __call__(self, image):
            f, b = [], []
            for i in range(image.shape[0]):
                n = 1 + int(float(image[i, 0, 0, :8].abs().sum()) * 7) % 20
                seed = int(float(image[i, 1, 1, :8].abs().sum()) * 1000) % 97
                fs, bs = _objects([n], 200 + seed, H, W)
                f.append(fs[0].to(image.device))
                b.append(bs[0].to(image.device))
            return f, b, None

    m, _, args = _model(dict(strategy="learned_bbox_wh", use_2_saca=True), H, W, 29, provider=Prov())
    imgs = [gen.randn(f"im{i}", (B, 3, H, W), 300 + i).cuda() for i in range(N)]
    gts = [(torch.rand(B, 1, H, W, generator=torch.Generator().manual_seed(i)) * 9.0 + 0.5).cuda() for i in range(N)]
    seq = ValidationStep(m, args, joint=True)
    ref = torch.cat([seq(imgs[i], gts[i], first_image_id=i)[0] for i in range(N)], 0)
    pv = PipelinedValidation(m, args, imgs[0], object_capacity=24)
    assert len(pv.graphs) == 4
    for i in range(N):
        pv.submit(imgs[i], gts[i], first_image_id=i)
    rec = pv.collect()
    assert rec.shape == (N, 10) and torch.equal(rec[:, 8:], ref[:, 8:])                 # counts and image ids exact
    assert rel_dev(rec[:, :8], ref[:, :8]) < 1e-5                                        # (capacity 24 vs the pair's own Nmax: rounding)
    assert pv.collect().shape[0] == 0
