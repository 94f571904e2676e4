"""-m "not gpu": host-side logic of the product package (no kernels run)."""
import ctypes
import os
import re

import pytest
import torch

from objcavit_amd import _lib
from objcavit_amd.config import AttrDict, load_reference_config, load_yaml, make_args
from util import load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_and_binding_declare_the_same_symbols():
    hdr = open(os.path.join(ROOT, "include", "objcavit_hip.h")).read()
    declared = set(re.findall(r"\b(ocv_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)


def test_library_loads_and_exports_every_symbol():
    """The C-ABI library must be present (built by __graft_entry__.build()) and export the whole header.
    No compute call is made: there is no GPU here."""
    if not os.path.exists(_lib.LIB_PATH):
        from objcavit_amd.build import build
        build()
    lib = _lib.load()
    for name in _lib.PROTOTYPES:
        assert hasattr(lib, name), name
    assert lib.ocv_abi_version() == _lib.ABI_VERSION == 5
    # argument validation runs on the host before any launch
    assert lib.ocv_patch_embed_workspace_bytes(16, 128, 240, 320, 128) == 16 * 4800 * 128 * 4      # 16 K slices x [M, E] fp32
    assert lib.ocv_patch_embed_workspace_bytes(1, 128, 240, 320, 64) == 0
    assert lib.ocv_bin_head_workspace_bytes(2, 256, 128) == 2 * 256 * 128 * 4
    rc = lib.ocv_linear_fwd(None, 4, 0, None, 4, 0, 0, None, None, 4, 0, 1, 1, 1, 4, 0, None)
    assert rc == -1 and b"null pointer" in lib.ocv_last_error()
    # dispatch switches are entry points, not environment reads inside the library
    assert lib.ocv_attention_set_dispatch(2) == -1 and b"form must be 0" in lib.ocv_last_error()
    assert lib.ocv_attention_set_dispatch(1) == 0 and lib.ocv_attention_set_dispatch(0) == 0
    csrc = os.path.join(ROOT, "objcavit_amd", "csrc")
    assert not any("getenv" in open(os.path.join(csrc, f)).read() for f in os.listdir(csrc)), "the C side reads no environment variable"


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(_lib.HipLibraryError):
        _lib.load(str(tmp_path / "nope.so"))


def test_cpu_tensors_are_rejected_not_silently_computed():
    from objcavit_amd import hip_ops
    with pytest.raises(_lib.HipLibraryError):
        hip_ops.linear(torch.zeros(4, 8), torch.zeros(3, 8))
    from objcavit_amd.modules.miniViT import mViT
    m = mViT(128).eval()
    with pytest.raises(_lib.HipLibraryError):
        m(torch.zeros(1, 128, 176, 192))


def test_attrdict_and_make_args():
    a = make_args(strategy="learned_bbox_wh", use_2_saca=True, dataset="kitti")
    assert a.graphbins.objcavit.positional_embedding_strategy == "learned_bbox_wh"
    assert a[a.model.name].objcavit.get("use_2_saca") is True
    assert a.graphbins.objcavit.get("no_obj_sa") is None
    assert a[a.basic.dataset].max_depth == 80
    d = AttrDict(x={"y": [1, {"z": 2}]})
    assert d.x.y[1].z == 2


def test_reference_yaml_files_parse():
    ref = "/root/reference/params/nyu_adabins_enet-b5.yaml"
    if not os.path.exists(ref):
        pytest.skip("reference tree not present")
    a = load_yaml(ref)
    assert a.model.name == "adabins" and a.adabins.n_bins == 256 and "efficientnet-b5" in a.adabins.encoder_name


@pytest.mark.parametrize("fixture,build", [
    ("g1_mvit_mini", lambda: __import__("objcavit_amd.modules.miniViT", fromlist=["mViT"]).mViT(128)),
    ("g2_saca_16_5", lambda: __import__("objcavit_amd.modules.ObjCAViT", fromlist=["x"]).SelfAttnCrossAttn(make_args(), 128, 4, 1024)),
    ("g2_saca_16_5_nosa", lambda: __import__("objcavit_amd.modules.ObjCAViT", fromlist=["x"]).SelfAttnCrossAttn(make_args(no_obj_sa=True), 128, 4, 1024)),
])
def test_state_dict_keys_match_reference(fixture, build):
    """Checkpoint interchange: the drop-in exposes exactly the reference's keys and shapes."""
    meta, _ = load_golden(fixture)
    sd = build().state_dict()
    assert {k: list(v.shape) for k, v in sd.items()} == meta["shapes"]


def test_full_model_state_dict_keys_match_reference():
    from objcavit_amd.modules.AdaBins import AdaBins
    from objcavit_amd.modules.ObjCAViT import ObjCAViT
    meta, _ = load_golden("g5_adabins_mini")
    sd = AdaBins(make_args(model="adabins")).state_dict()
    assert {k: list(v.shape) for k, v in sd.items()} == meta["shapes"]
    for tag, kw in (("learned", {}), ("bbox_wh_2saca", dict(strategy="learned_bbox_wh", use_2_saca=True)),
                    ("grid_random", dict(strategy="grid_random"))):
        meta, _ = load_golden(f"g3_objcavit_{tag}")
        a = make_args(dimensions_train=[352, 384], dimensions_test=[352, 384], **kw)
        sd = ObjCAViT(a, embedding_dim=128, max_seq_len=500).state_dict()
        assert {k: list(v.shape) for k, v in sd.items()} == meta["shapes"], tag


# ------------------------------------------------------------------ row N3: checkpoint / config ingestion
def test_lightning_checkpoint_ingestion(tmp_path):
    """A checkpoint laid out like the reference's Lightning .ckpt (model.* keys + metric / loss states) loads into the
    drop-in by key; a wrong shape or a missing key is reported, not swallowed."""
    import torch
    from objcavit_amd.checkpoint import extract_model_state, load_reference_checkpoint
    from objcavit_amd.modules.miniViT import mViT
    src, dst = mViT(128), mViT(128)
    with torch.no_grad():
        for i, p in enumerate(src.parameters()):
            p.copy_(torch.full_like(p, 0.01 * (i + 1)))
    lightning_sd = {"model." + k: v.clone() for k, v in src.state_dict().items()}
    lightning_sd.update({"abs_rel.normed_abs_diff_total": torch.zeros(1), "abs_rel_ra.batch_count": torch.tensor(0),
                         "loss.some_buffer": torch.ones(2)})
    ckpt = {"state_dict": lightning_sd, "epoch": 3, "global_step": 100, "pytorch-lightning_version": "1.7.0"}
    path = tmp_path / "last.ckpt"
    torch.save(ckpt, path)
    missing, unexpected = load_reference_checkpoint(dst, str(path))
    assert not missing and not unexpected
    for (k, a), (_, b) in zip(src.state_dict().items(), dst.state_dict().items()):
        assert torch.equal(a, b), k
    # bare state dicts pass through; prototype-layer keys survive the prefix strip
    bare = extract_model_state(src.state_dict())
    assert set(bare) == set(src.state_dict())
    # damage is reported
    bad = dict(lightning_sd)
    bad.pop("model." + next(iter(src.state_dict())))
    with pytest.raises(RuntimeError, match="missing"):
        load_reference_checkpoint(mViT(128), {"state_dict": bad})
    k0 = "model." + next(iter(src.state_dict()))
    bad = dict(lightning_sd)
    bad[k0] = torch.zeros(3)
    with pytest.raises(RuntimeError, match="shapes"):
        load_reference_checkpoint(mViT(128), {"state_dict": bad})


def test_every_reference_params_file_builds_args():
    """All of the reference's params/*.yaml parse without OmegaConf and carry the knobs the hot path reads."""
    import glob
    files = sorted(glob.glob("/root/reference/params/*graphbins*.yaml") + glob.glob("/root/reference/params/*adabins*.yaml"))
    if not files:
        pytest.skip("reference tree not present")
    import yaml
    malformed = []
    for f in files:
        try:
            a = load_reference_config(f) if len(malformed) % 2 == 0 else load_reference_config(f, "/root/reference/params/basicParams.yaml")
        except yaml.YAMLError:                # a stray "." line in one of the reference's files: not valid YAML for anyone
            malformed.append(os.path.basename(f))
            continue
        blk = a[a.model.name]
        assert a.model.name in ("graphbins", "adabins") and a.basic.dataset in ("nyu", "kitti"), f
        assert blk.n_bins == 256 and "efficientnet" in blk.encoder_name, f
        ds = a[a.basic.dataset]
        assert ds.min_depth > 0 and ds.max_depth in (10, 80) and len(ds.dimensions_test) == 2, f
        if a.model.name == "graphbins":
            oc = blk.objcavit
            assert oc.positional_embedding_strategy in ("learned", "learned_bbox_wh", "grid_random", "grid_random_roi_align"), f
            assert oc.embedding_dim == 128, f
    assert len(malformed) <= 2 and len(files) - len(malformed) >= 10, malformed
    # the built-in dataset constants are basicParams.yaml's
    f = files[0] if os.path.basename(files[0]) not in malformed else files[-1]
    a, b = load_reference_config(f), load_reference_config(f, "/root/reference/params/basicParams.yaml")
    for ds in ("nyu", "kitti"):
        for k in ("min_depth", "max_depth", "dimensions_test", "eigen_crop", "garg_crop", "do_kb_crop"):
            assert a[ds][k] == b[ds][k], (ds, k)


# ------------------------------------------------------------------ row N4: object front-end formats
def test_relative_size_index_matches_hand_computed_cases():
    import torch
    from objcavit_amd.objects import REL_SIZE_SCALE, relative_size_index
    assert len(REL_SIZE_SCALE) == 7
    assert relative_size_index(None) == [] and relative_size_index(torch.tensor([[1., 1., 5., 5.]])) == []
    # areas 100, 100, 1, 10000: equal -> log 0 -> (0+1)/2*4 = 2 -> index 3 ("about the same size as")
    boxes = torch.tensor([[0., 0., 10., 10.], [0., 0., 20., 5.], [0., 0., 1., 1.], [0., 0., 100., 100.]])
    idx = relative_size_index(boxes)
    assert idx[0] == 3
    # 100 vs 1: log(100) = 4.6 -> (5.6/2)*4 = 11.2 -> 12 -> clipped to 6 ("much bigger than")
    assert idx[1] == 6
    # 1 vs 10000: log = -9.2 -> negative -> clipped to 0 ("much smaller than"); 10000 vs 100 (cyclic) -> 6
    assert idx[2] == 0 and idx[3] == 6
    # just inside the scale: ratio e -> (1+1)/2*4 = 4 -> 5 ("bigger than"); ratio 1/e -> 0 -> 1 ("smaller than")
    import math
    b = torch.tensor([[0., 0., math.e, 1.], [0., 0., 1., 1.]], dtype=torch.float64)
    assert relative_size_index(b) == [5, 1]


def test_table_object_provider_formats():
    import torch
    from objcavit_amd.objects import TableObjectProvider
    table = torch.arange(6 * 512, dtype=torch.float32).reshape(6, 512)
    dets = ([torch.tensor([[10., 20., 30., 40.], [50., 60., 8., 8.]]), None, torch.tensor([[1., 2., 3., 4.]])],
            [torch.tensor([5, 2]), None, [0]])
    prov = TableObjectProvider(lambda img: dets, class_table=table)
    feats, boxes, ann = prov(torch.zeros(3, 3, 8, 8))
    assert ann is None and [f.shape for f in feats] == [(2, 512), (1, 512), (1, 512)]
    assert torch.equal(feats[0], table[[5, 2]]) and float(feats[1].abs().sum()) == 0.0 and boxes[1] is None
    assert torch.equal(boxes[0], dets[0][0]) and feats[2].dtype == torch.float32
    # relative-size strategy: the cache is asked for (class, next class, relation)
    asked = []

    def phrase(c, cn, rel):
        asked.append((c, cn, rel))
        return torch.full((512,), float(c * 100 + cn * 10 + rel))

    prov = TableObjectProvider(lambda img: dets, phrase_features=phrase)
    feats, boxes, _ = prov(torch.zeros(3, 3, 8, 8))
    assert asked == [(5, 2, 6), (2, 5, 0), (0, 0, -1)]          # areas 1200 vs 64; a lone object has no relation
    assert float(feats[0][0, 0]) == 526.0 and float(feats[0][1, 0]) == 250.0
    with pytest.raises(ValueError):
        TableObjectProvider(lambda img: dets)
    with pytest.raises(ValueError):
        TableObjectProvider(lambda img: ([torch.zeros(1, 4)], [torch.tensor([9])]), class_table=table)(torch.zeros(1, 3, 8, 8))


def test_relative_size_index_vs_reference_golden():
    """Row N4: objects.relative_size_index against the relation indices produced by the reference's own
    ObjectLanguageStrategy.get_single_relative_size_clause (tests/golden/g7_relsize.npz): random boxes, equal areas, the
    half-way points of the rounding, both ends of the 7-point scale, a single object, no detections."""
    from objcavit_amd import objects
    meta, z = load_golden("g7_relsize")
    assert tuple(meta["scale"]) == objects.REL_SIZE_SCALE
    seen = set()
    for i in range(meta["n_images"]):
        xywh = None if meta["none"][i] else torch.from_numpy(z[f"xywh{i}"])
        got = objects.relative_size_index(xywh)
        assert got == z[f"idx{i}"].tolist(), i
        seen.update(got)
        clauses = [c for c in meta["clauses"][i] if c]
        assert len(clauses) == len(got)
        for c, r in zip(clauses, got):
            assert f" appears {objects.REL_SIZE_SCALE[r]} the " in c          # the phrase the reference handed to CLIP
    assert seen == set(range(7))                                              # every entry of the scale is exercised


def test_encoder_layer_params_carry_their_size():
    """ABI 2: ocv_encoder_layer_params starts with struct_size; the library reads only that many bytes (later fields =
    NULL) and rejects a size that cannot hold the twelve fp32 parameter pointers -- checked on the host, before any launch."""
    import ctypes as C
    lib = _lib.load()
    p = _lib.EncoderLayerParams()
    assert p.struct_size == C.sizeof(_lib.EncoderLayerParams) == 8 * 21       # size + 12 fp32 + 4 three-term + 4 two-term fp16 pointers
    one = C.c_void_p(256)                                   # non-null, never dereferenced: validation fails first
    for bad in (0, 8 * 12, 8 * 13 + 4, 1 << 20):
        p.struct_size = bad
        rc = lib.ocv_encoder_layer_fwd(one, C.byref(p), None, 0, one, 1, 8, 128, 4, 1024, 1e-5, one, 1 << 30, None)
        assert rc == -1 and b"struct_size" in lib.ocv_last_error(), bad
        rc = lib.ocv_encoder_stack_fwd(one, C.byref(p), 1, None, 0, one, 1, 8, 128, 4, 1024, 1e-5, one, 1 << 30, None)
        assert rc == -1 and b"struct_size" in lib.ocv_last_error(), bad
    # a caller built against a header WITHOUT the packed-weight fields (size + 12 pointers): accepted, *_p3 read as NULL,
    # which ocv_encoder_stack_fwd then reports as "lacks its packed split3 weights" instead of reading garbage
    p.struct_size = 8 * 13
    p.in_proj_p3 = 0xdead0                                   # beyond struct_size: must be ignored
    rc = lib.ocv_encoder_stack_fwd(one, C.byref(p), 1, None, 0, one, 1, 8, 128, 4, 1024, 1e-5, one, 1 << 30, None)
    assert rc == -1 and b"lacks its packed split3 weights" in lib.ocv_last_error()


def test_pad_objects_to_a_larger_batch_nmax():
    """SelfAttnCrossAttn._pad_objects(pad_to=...): rows beyond a list's count are 1e-4 and masked, up to the agreed Nmax."""
    from objcavit_amd.modules.ObjCAViT import PAD_VALUE, SelfAttnCrossAttn
    objs = [torch.ones(3, 4), torch.ones(2, 4) * 2]
    f, m = SelfAttnCrossAttn._pad_objects(objs, torch.device("cpu"))
    assert f.shape == (2, 3, 4) and m.tolist() == [[False, False, False], [False, False, True]]
    f, m = SelfAttnCrossAttn._pad_objects(objs, torch.device("cpu"), pad_to=5)
    assert f.shape == (2, 5, 4) and m.tolist() == [[False] * 3 + [True] * 2, [False] * 2 + [True] * 3]
    assert float(f[0, 3:].max()) == float(f[1, 2:].min()) == float(torch.tensor(PAD_VALUE)) and float(f[1, 1, 0]) == 2.0
    f, m = SelfAttnCrossAttn._pad_objects([torch.ones(2, 4), torch.ones(2, 4)], torch.device("cpu"), pad_to=4)   # equal counts, still padded
    assert f.shape == (2, 4, 4) and m.sum().item() == 4
    with pytest.raises(ValueError):
        SelfAttnCrossAttn._pad_objects(objs, torch.device("cpu"), pad_to=2)


def test_island_hook_and_workspace_scope_are_thread_local():
    """The capturer's island hook and the workspace stack are per-thread context objects (not module globals): while one
    thread holds an island scope open, a launch of the SAME name on another thread is not diverted into it, and a
    workspace scope entered on one thread is invisible to the other.  (No GPU call: the launches are plain closures.)"""
    import threading
    from objcavit_amd import hip_ops
    seen = {"diverted": [], "ran": []}
    hook = hip_ops.IslandHook(("conv3x3|x",), lambda name, call: seen["diverted"].append(name))
    store = hip_ops.WorkspaceStore()
    entered, release = threading.Event(), threading.Event()
    err = []

    def holder():
        try:
            with hip_ops.island_scope(hook), hip_ops.workspace_scope(store):
                hip_ops.launch("conv3x3|x", lambda: seen["ran"].append("holder"))          # -> the hook
                hip_ops.launch("other", lambda: seen["ran"].append("holder-other"))        # -> runs
                assert hip_ops._ws_stack()[-1] is store
                entered.set()
                release.wait(timeout=30)
        except Exception as e:          # noqa: BLE001
            err.append(e)
            entered.set()

    t = threading.Thread(target=holder)
    t.start()
    assert entered.wait(timeout=30)
    hip_ops.launch("conv3x3|x", lambda: seen["ran"].append("main"))                         # other thread: NOT diverted
    assert hip_ops._ws_stack()[-1] is not store
    with pytest.raises(RuntimeError):
        with hip_ops.island_scope(hook), hip_ops.island_scope(hook):                         # one capture per thread at a time
            pass
    release.set()
    t.join()
    assert not err, err
    assert seen["diverted"] == ["conv3x3|x"] and seen["ran"] == ["holder-other", "main"]


def test_padded_objects_from_lists_and_weight_preparation_on_the_host():
    """Round 4 host logic without a GPU: the reference's two object lists -> padded tensors + int32 counts (an image without
    detections keeps ONE row with the box (-1, -1, -1, -1)); fp16 weight pairs scaled per output channel by exact powers of two;
    the fp16-fit test of a weight matrix; the Winograd F(4x4,3x3) filters' per-position and per-channel powers of two."""
    from objcavit_amd import hip_ops
    from objcavit_amd.modules.ObjCAViT import PaddedObjects
    feats = [torch.ones(3, 512), torch.full((1, 512), 2.0), torch.full((5, 512), 3.0)]
    boxes = [torch.rand(3, 4), None, torch.rand(5, 4)]
    po = PaddedObjects.from_lists(feats, boxes, torch.device("cpu"), capacity=8)
    assert po.features.shape == (3, 8, 512) and po.xywh.shape == (3, 8, 4) and po.counts.dtype == torch.int32
    assert po.counts.tolist() == [3, 1, 5] and po.max_count == 5 and po.capacity == 8
    assert po.xywh[1, 0].tolist() == [-1.0] * 4 and float(po.features[2, 4, 0]) == 3.0 and float(po.features[0, 3:].abs().max()) == 0.0
    with pytest.raises(ValueError):
        PaddedObjects.from_lists([torch.ones(2, 512)], [torch.rand(3, 4)], torch.device("cpu"))       # counts disagree
    with pytest.raises(ValueError):
        PaddedObjects.from_lists([torch.ones(0, 512)], [torch.rand(0, 4)], torch.device("cpu"))       # no row at all

    w = torch.randn(24, 40, 3, 3) * torch.logspace(-3, 3, 24).view(24, 1, 1, 1)                      # heavy-tailed OUTPUT channels
    hi, lo, osc = hip_ops.prep_conv_weight(w, f16=True)
    assert hi.dtype == lo.dtype == torch.float16 and hi.shape == (9, 24, 64) and osc.shape == (24,)
    assert bool((torch.log2(osc) == torch.round(torch.log2(osc))).all())                             # exact powers of two
    rowmax = hi.float().abs().amax(dim=(0, 2))
    assert float(rowmax.min()) >= 2.0 ** 7.4 and float(rowmax.max()) <= 2.0 ** 8.6                   # every row's largest entry near 2^8
    back = (hi.float() + lo.float()) * osc.view(1, 24, 1)
    ref = w.permute(2, 3, 0, 1).reshape(9, 24, 40)
    assert float(((back[:, :, :40] - ref).abs() / ref.abs().amax(dim=(0, 2), keepdim=True)).max()) < 2.0 ** -21   # 22-bit pairs per row
    assert not bool(back[:, :, 40:].any())
    bh, bl = hip_ops.prep_conv_weight(w)                                                               # bf16 pairs: unchanged contract
    assert bh.dtype == torch.bfloat16 and bh.shape == (9, 24, 64)

    flat = torch.randn(16, 72)
    assert hip_ops.fp16_weight_safe(flat)
    flat[:, 5] *= 2.0 ** -20
    assert not hip_ops.fp16_weight_safe(flat)
    flat[:, 5] = 0.0
    assert hip_ops.fp16_weight_safe(flat)                                                              # an all-zero column is fine

    span = torch.logspace(-3, 3, 40).view(1, 40, 1, 1)
    u_hi, u_lo, fs, cs = hip_ops.prep_winograd43_weight(torch.randn(24, 40, 3, 3) / span)
    assert u_hi.shape == (36, 24, 64) and fs.shape == (36,) and cs.shape == (64,) and bool((cs[40:] == 1).all())
    assert bool((torch.log2(cs) == torch.round(torch.log2(cs))).all()) and float(cs.max()) == 1.0
    colmax = u_hi.float().abs().amax(dim=(0, 1))[:40]
    assert float(colmax.max() / colmax.min()) < 8.0                                                    # columns equalised (were 10^6 apart)


def test_validation_step_joint_forward_on_a_stand_in_model():
    """ValidationStep(joint=True) on a model that declares its images independent: ONE call on [batch | mirrored batch], the
    un-mirrored half returned, the mirrored depth handed to the metric kernel (checked here up to the kernel call)."""
    import collections
    from objcavit_amd.config import make_args
    from objcavit_amd.validation import ValidationStep
    Out = collections.namedtuple("Out", ["depth_pred", "bin_edges"])
    calls = []

    class Toy(torch.nn.Module):
        images_are_independent = True

        def forward(self, image):
            calls.append(tuple(image.shape))
            return Out(image[:, :1, ::2, ::2].abs() + torch.linspace(0.5, 4.0, image.shape[3] // 2), image.mean(dim=(1, 2, 3), keepdim=False)[:, None])

    img = torch.randn(3, 3, 8, 12)
    vs = ValidationStep(Toy(), make_args(), joint=True)
    out, mirror = vs._forward_pair(img)
    assert calls == [(6, 3, 8, 12)] and out.depth_pred.shape == (3, 1, 4, 6) and mirror.shape == (3, 1, 4, 6) and out.bin_edges.shape == (3, 1)
    two = ValidationStep(Toy(), make_args(), joint=False)
    o2, m2 = two._forward_pair(img)
    assert torch.equal(out.depth_pred, o2.depth_pred) and torch.equal(mirror, m2)


def test_validation_step_on_a_captured_graph_stand_in():
    """a captured graph has ``__call__`` and ``static_image`` but no ``forward``.  Captured for B images it serves the
    image / mirror pair as two replays (and the first replay's STATIC result tensors are copied before the second overwrites them);
    captured for 2B images with ``object_group = B`` it takes the joint forward."""
    import collections
    from objcavit_amd.config import make_args
    from objcavit_amd.validation import ValidationStep, _takes_group
    Out = collections.namedtuple("Out", ["depth_pred", "bin_edges", "detections"])

    class Graphish:
        images_are_independent = True

        def __init__(self, n, group=None):
            self.static_image = torch.zeros(n, 3, 8, 12)
            self.object_group = group
            self.calls = []
            self.edges = torch.zeros(n, 1)                      # static, overwritten by every "replay"

        def __call__(self, image, object_features=None, object_xywh_list=None):
            assert image.shape == self.static_image.shape
            self.calls.append(tuple(image.shape))
            self.edges.copy_(image.mean(dim=(1, 2, 3))[:, None])
            return Out(image[:, :1, ::2, ::2].abs() + 0.5, self.edges, None)

    img = torch.randn(3, 3, 8, 12)
    g = Graphish(3)
    assert not _takes_group(g)
    out, mirror = ValidationStep(g, make_args(), joint=True)._forward_pair(img)
    assert g.calls == [(3, 3, 8, 12)] * 2
    assert torch.equal(out.bin_edges, img.mean(dim=(1, 2, 3))[:, None])          # not the mirror's replay
    assert torch.equal(mirror, img.flip(dims=[3])[:, :1, ::2, ::2].abs() + 0.5)
    g2 = Graphish(6, group=3)
    out2, mirror2 = ValidationStep(g2, make_args(), joint=True)._forward_pair(img)
    assert g2.calls == [(6, 3, 8, 12)] and torch.equal(out2.depth_pred, out.depth_pred) and torch.equal(mirror2, mirror)
    g3 = Graphish(6, group=None)                                                  # 2B images but not grouped: not a joint capture of B
    with pytest.raises(AssertionError):
        ValidationStep(g3, make_args(), joint=True)._forward_pair(img)


def test_side_stream_switches_follow_the_batches_in_flight(monkeypatch):
    """The forward's four forks under ONE switch, OCV_FORKS: 'auto' (default) = on for a lone batch on at most four hardware queues,
    off inside ``hip_ops.batches_in_flight(n > 1)`` (thread-local: another thread's owner is not affected) and off when the process
    asked for more than four hardware queues (profiles/r05_graph_shapes.txt: a forked graph replays 3x slower there); '0' / '1'
    override; anything else is an error, not a silent default; ``hip_ops.forks(name=...)`` forces single forks (thread-local) and
    wins over the environment; inside hip_ops.single_chain() no further fork is offered; launches inside islands_suspended() stay in
    the capture."""
    import threading
    from objcavit_amd import hip_ops
    switches = {"obj": hip_ops.object_prepass_enabled, "token": hip_ops.token_overlap_enabled,
                "head": hip_ops.head_overlap_enabled, "skip": hip_ops.skip_overlap_enabled}
    monkeypatch.delenv("OCV_FORKS", raising=False)
    monkeypatch.delenv("GPU_MAX_HW_QUEUES", raising=False)
    assert all(f() for f in switches.values())
    with hip_ops.batches_in_flight(3):
        assert not any(f() for f in switches.values())
        seen = []
        t = threading.Thread(target=lambda: seen.append(all(f() for f in switches.values())))      # another owner, another thread
        t.start()
        t.join()
        assert seen == [True]
        with hip_ops.batches_in_flight(0):                                                         # (clamped to 1), nested
            assert all(f() for f in switches.values())
        assert not any(f() for f in switches.values())
        monkeypatch.setenv("OCV_FORKS", "1")
        assert all(f() for f in switches.values())
        monkeypatch.setenv("OCV_FORKS", "0")
        assert not any(f() for f in switches.values())
        monkeypatch.setenv("OCV_FORKS", "yes")
        for f in switches.values():
            with pytest.raises(ValueError):
                f()
        monkeypatch.delenv("OCV_FORKS")
        for k, f in switches.items():                                                              # one fork forced, the others follow 'auto'
            with hip_ops.forks(**{k: True}):
                assert f() and sum(g() for g in switches.values()) == 1
    assert all(f() for f in switches.values())
    with hip_ops.forks(head=False, skip=False):
        assert hip_ops.object_prepass_enabled() and not hip_ops.head_overlap_enabled() and not hip_ops.skip_overlap_enabled()
        with hip_ops.forks(head=True):
            assert hip_ops.head_overlap_enabled() and not hip_ops.skip_overlap_enabled()
        assert not hip_ops.head_overlap_enabled()
    with pytest.raises(ValueError):
        hip_ops.forks(decoder=True)
    for q, ok in (("2", True), ("4", True), ("6", False), ("8", False), ("16", False)):
        monkeypatch.setenv("GPU_MAX_HW_QUEUES", q)
        assert hip_ops.hw_queues_allow_forks() == ok and all(f() == ok for f in switches.values()), q
    with hip_ops.forks(skip=True):                                                                 # an explicit request is obeyed
        assert hip_ops.skip_overlap_enabled()
    monkeypatch.delenv("GPU_MAX_HW_QUEUES")
    with hip_ops.single_chain():
        assert not hip_ops.token_overlap_enabled() and hip_ops.head_overlap_enabled()
    assert hip_ops.token_overlap_enabled()
    broke = []
    with hip_ops.island_scope(hip_ops.IslandHook(("x",), lambda name, call: broke.append(name))):
        with hip_ops.islands_suspended():
            hip_ops.launch("x", lambda: broke.append("ran"))
        hip_ops.launch("x", lambda: broke.append("never"))
    assert broke == ["ran", "x"]


def test_map_placeholder_and_fp32_map_on_the_host():
    """hip_ops.map_placeholder / fp32_map (the decoder's output in split form only): the placeholder has the map's shape, no storage
    and NaN contents; fp32_map rebuilds hi + lo from the hl32 layout (hi | lo per 32 channels, pad channels dropped), returns real
    tensors unchanged, and keeps the split copy attached."""
    from objcavit_amd import hip_ops
    B, C, H, W = 2, 40, 3, 5                                   # 40 channels: two 32-blocks, the second one padded
    x = torch.randn(B, C, H, W)
    Cp = 64
    xp = torch.zeros(B, H, W, Cp)
    xp[..., :C] = x.permute(0, 2, 3, 1)
    for dt, tol in ((torch.float16, 2.0 ** -21), (torch.bfloat16, 2.0 ** -15)):
        hi = xp.to(dt)
        lo = (xp - hi.float()).to(dt)
        hl = torch.stack([hi.view(B, H, W, Cp // 32, 32), lo.view(B, H, W, Cp // 32, 32)], dim=4).reshape(B, H, W, 2 * Cp)
        sp = hip_ops.SplitAct(hl.contiguous(), C)
        ph = hip_ops.map_placeholder(sp)
        assert tuple(ph.shape) == (B, C, H, W) and ph.stride() == (0, 0, 0, 0) and bool(torch.isnan(ph).all())
        assert ph._ocv_fp32_missing and ph._ocv_split is sp
        full = hip_ops.fp32_map(ph)
        assert tuple(full.shape) == (B, C, H, W) and full.is_contiguous(memory_format=torch.channels_last) and full._ocv_split is sp
        assert float((full - x).abs().max() / x.abs().max()) < tol
    real = torch.randn(1, 8, 2, 2)
    assert hip_ops.fp32_map(real) is real
