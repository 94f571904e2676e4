"""-m "not gpu": the CPU oracle (oracle/restate.py) against the golden vectors
that tests/golden/make_golden.py produced by running the reference's own
modules.  These pin the oracle; the GPU tests then compare the HIP path with
the oracle and with the same fixtures."""
import numpy as np
import pytest
import torch

import gen
from oracle import restate
from util import gains_of, load_golden, rel_dev, state_dict_from

torch.set_grad_enabled(False)
TOL = 3e-5


@pytest.mark.parametrize("tag", ["mini", "nyu"])
def test_g1_mvit(tag):
    meta, z = load_golden(f"g1_mvit_{tag}")
    sd = state_dict_from(meta["shapes"], meta["seed"], gains_of(meta))
    x = gen.randn("x", (meta["B"], 128, meta["fh"], meta["fw"]), meta["seed"])
    y, ram = restate.mvit_forward(x, sd)
    tgt = restate.patch_transformer_encoder(x, sd, "patch_transformer.")
    assert rel_dev(y, z["y"]) < TOL
    assert rel_dev(ram.flatten(2)[:, :, torch.from_numpy(z["pix"])], z["ram_px"]) < TOL
    assert rel_dev(tgt, z["tgt"]) < TOL
    assert abs(float(y.sum(1).mean()) - 1.0) < 1e-5      # widths are normalised


@pytest.mark.parametrize("tag", ["16_5", "1_1", "100_3", "16_5_nosa", "8_8_8"])
def test_g2_saca(tag):
    meta, z = load_golden(f"g2_saca_{tag}")
    sd = state_dict_from(meta["shapes"], meta["seed"], gains_of(meta))
    S, E = meta["S"], 128
    tok = gen.randn("tok", (len(meta["counts"]), S, E), meta["seed"])
    objs = [gen.randn(f"obj{i}", (n, E), meta["seed"]) for i, n in enumerate(meta["counts"])]
    fi, fo, inter = restate.saca_forward(tok, objs, sd, "", no_obj_sa=meta["no_obj_sa"])
    assert rel_dev(fi, z["final_img"]) < TOL
    assert rel_dev(fo, z["final_obj"]) < TOL
    if not meta["no_obj_sa"]:
        # padded object rows leave the encoder as exact zeros
        for b, n in enumerate(meta["counts"]):
            assert float(inter["att_obj"][b, n:].abs().max() if n < inter["att_obj"].shape[1] else 0.0) == 0.0


def test_q1_degenerate_cross_attention():
    """with Nmax <= S/2 every image token receives the same vector."""
    meta, z = load_golden("g2_saca_16_5")
    fi = torch.from_numpy(z["final_img"])
    assert float((fi - fi[:, :1]).abs().max()) < 1e-6
    meta, z = load_golden("g2_saca_100_3")
    fi = torch.from_numpy(z["final_img"])
    assert float((fi[0] - fi[0, :1]).abs().max()) > 1e-3      # Nmax > S/2: real objects are attended


def _objcavit_inputs(meta):
    fh, fw, seed = meta["fh"], meta["fw"], meta["seed"]
    H, W = 2 * fh, 2 * fw
    x = gen.randn("x", (len(meta["counts"]), 128, fh, fw), seed)
    feats, xywh = [], []
    for i, n in enumerate(meta["counts"]):
        k = 1 if n is None else n
        feats.append(gen.randn(f"f{i}", (k, 512), seed, 10.0 / np.sqrt(512)))
        xywh.append(None if n is None else gen.boxes(f"b{i}", n, seed, H, W))
    return x, feats, xywh


G3_TAGS = ["learned", "learned_nosa", "bbox_wh_2saca", "learned_2saca_eq", "grid_random", "learned_many",
           "bbox_wh_2saca_many"]


@pytest.mark.parametrize("tag", G3_TAGS)
def test_g3_objcavit(tag):
    meta, z = load_golden(f"g3_objcavit_{tag}")
    sd = state_dict_from(meta["shapes"], meta["seed"], gains_of(meta))
    x, feats, xywh = _objcavit_inputs(meta)
    y, ram, inter = restate.objcavit_forward(x, feats, xywh, sd, "", return_intermediates=True, **meta["kw"])
    assert rel_dev(y, z["y"]) < TOL
    assert rel_dev(ram.flatten(2)[:, :, torch.from_numpy(z["pix"])], z["ram_px"]) < TOL
    assert rel_dev(inter["saca1_img"], z["saca1_img"]) < TOL


def test_g4_decoder_and_encoder_order():
    from oracle import effnet_ref
    meta, z = load_golden("g4_decoder")
    sd_e = state_dict_from(meta["shapes_enc"], meta["seed"])
    sd_d = state_dict_from(meta["shapes_dec"], meta["seed"])
    x = gen.randn("img", (1, 3, 96, 128), meta["seed"])
    feats = effnet_ref.encoder_features(x, sd_e, "original_model.")
    assert [int(f.shape[1]) for f in feats] == meta["chans"]
    assert np.allclose([float(feats[i].mean()) for i in (4, 5, 6, 8, 11)], z["skip_means"], rtol=1e-4, atol=1e-6)
    out = restate.decoder_forward(feats, sd_d, "")
    assert rel_dev(out, z["out"]) < TOL


@pytest.mark.parametrize("tag", ["mini", "nyu"])
def test_g5_adabins_config1(tag):
    """BASELINE configs[0]: AdaBins enet-b5 NYU, CPU forward."""
    meta, z = load_golden(f"g5_adabins_{tag}")
    sd = state_dict_from(meta["shapes"], meta["seed"], gains_of(meta))
    img = gen.randn("img", (1, 3, meta["H"], meta["W"]), meta["seed"])
    depth, edges = restate.adabins_forward(img, sd, 0.001, 10)
    assert meta["fields"] == ["depth_pred", "bin_edges"]
    assert depth.shape == (1, 1, meta["H"] // 2, meta["W"] // 2) and edges.shape == (1, 257)
    ref = torch.from_numpy(z["depth_px"])
    got = depth.flatten()[torch.from_numpy(z["pix"])]
    assert float(((got - ref).abs() / ref).max()) < 1e-4
    assert rel_dev(edges, z["bin_edges"]) < TOL
    assert abs(float(depth.min()) - float(z["depth_stats"][0])) < 1e-3


def test_g5_adabins_final_upscale():
    """do_final_upscale (51 of the reference's 108 params files): the reference's own AdaBins with the decoder's fifth
    stage -- features, 12 x 13 patch grid and depth at FULL resolution, positional table of 1200 rows."""
    meta, z = load_golden("g5_adabins_mini_upscale")
    assert meta["do_final_upscale"] and meta["shapes"]["adaptive_bins_layer.patch_transformer.positional_encodings"] == [1200, 128]
    assert "dense_feature_extractor.decoder.final_upscale._net.0.weight" in meta["shapes"]
    sd = state_dict_from(meta["shapes"], meta["seed"], gains_of(meta))
    img = gen.randn("img", (1, 3, meta["H"], meta["W"]), meta["seed"])
    depth, edges = restate.adabins_forward(img, sd, 0.001, 10, do_final_upscale=True)
    assert depth.shape == (1, 1, meta["H"], meta["W"]) and edges.shape == (1, 257)
    ref = torch.from_numpy(z["depth_px"])
    got = depth.flatten()[torch.from_numpy(z["pix"])]
    assert float(((got - ref).abs() / ref).max()) < 1e-4
    assert rel_dev(edges, z["bin_edges"]) < TOL


def test_ps_roi_align_restatement_vs_hand_computed_boxes():
    """ps_roi_align is third-party and absent (parity unpinned): the restatement is checked against hand-computed
    boxes (tests/roi_cases.py) and against an independent separable float64 formulation on random boxes."""
    import roi_cases as rc
    table = rc.grid_table()
    grid = table.view(rc.GH, rc.GW, 2).permute(2, 0, 1).unsqueeze(0).contiguous()
    boxes = torch.tensor([c[1] for c in rc.CASES])
    got = restate.ps_roi_align_1x1(grid, restate._xywh_to_xyxy_clamped(boxes), rc.SCALE)
    want = torch.tensor([c[2] for c in rc.CASES], dtype=torch.float64)
    assert float((got.double() - want).abs().max()) < 1e-6, (got, want)
    nan = restate.ps_roi_align_1x1(grid, restate._xywh_to_xyxy_clamped(torch.tensor(rc.NAN_BOXES)), rc.SCALE)
    assert bool(torch.isnan(nan).all())
    # the module-level wrapper: "obj" space (scale 1 / (patch * factor)) and "img" space (scale 1 / patch)
    pos = restate.grid_random_pos_emb(table, boxes, (rc.GH * 16, rc.GW * 16), 16, "roi_align", "obj")
    assert float((pos.double() - want).abs().max()) < 1e-6
    pos_img = restate.grid_random_pos_emb(table, (boxes / 2)[None], (rc.GH * 16, rc.GW * 16), 16, "roi_align", "img")
    assert float((pos_img[0].double() - want).abs().max()) < 1e-6
    # random boxes on a random table, incl. boxes far larger than the image and boxes hanging over every edge
    rs = np.random.RandomState(5)
    gh, gw, E = 15, 20, 8
    tab = rs.uniform(0, 1, (gh * gw, E)).astype(np.float32)
    bx = np.stack([rs.uniform(-50, 700, 60), rs.uniform(-50, 530, 60), rs.uniform(0.5, 900, 60), rs.uniform(0.5, 700, 60)], 1)
    g2 = torch.from_numpy(tab).view(gh, gw, E).permute(2, 0, 1).unsqueeze(0).contiguous()
    got = restate.ps_roi_align_1x1(g2, restate._xywh_to_xyxy_clamped(torch.from_numpy(bx.astype(np.float32))), 1 / 32)
    want = rc.vectorised_expected(tab, gh, gw, bx, 1 / 32)
    assert float(np.abs(got.double().numpy() - want).max()) < 1e-6


def test_grid_sample_restatement_matches_torch():
    rs = np.random.RandomState(0)
    inp = torch.from_numpy(rs.standard_normal((2, 5, 7, 9)).astype(np.float32))
    grid = torch.from_numpy(rs.uniform(-1.3, 1.3, (2, 3, 11, 2)).astype(np.float32))
    ref = torch.nn.functional.grid_sample(inp, grid, mode="bilinear", padding_mode="zeros", align_corners=False)
    assert rel_dev(restate.grid_sample_bilinear_zeros(inp, grid), ref) < 1e-6


# ------------------------------------------------------------------ row N2: validation-step arithmetic
@pytest.mark.parametrize("tag", list(gen.VALIDATION_CASES))
def test_g6_validation_metrics(tag):
    """oracle/validation_ref.py against the numbers the reference's own MetricsPreprocess + metric classes produced."""
    from oracle import validation_ref as vr
    meta, z = load_golden(f"g6_validation_{tag}")
    gt, pa, pb = gen.validation_inputs(tag)
    rec = vr.per_image_records(pa, gt, meta["min_depth"], meta["max_depth"], meta["dataset"], meta["garg"], meta["eigen"],
                               depth_pred_mirror=pb)
    assert int(rec[:, 8].sum()) == int(z["n_valid"])
    tot = vr.batch_totals(rec)
    for i, k in enumerate(vr.METRICS):
        assert abs(tot[k] - float(z["metrics"][i])) <= 2e-6 * abs(float(z["metrics"][i])) + 1e-9, k
    assert rel_dev(rec, z["records"]) < 1e-6
    p, mask = vr.metrics_preprocess(vr.tta_average(pa, pb, meta["min_depth"], meta["max_depth"]), gt, meta["min_depth"],
                                    meta["max_depth"], meta["dataset"], meta["garg"], meta["eigen"])
    assert torch.equal(mask.sum((1, 3)).to(torch.int32), torch.from_numpy(z["mask_rows"]))
    assert rel_dev(p.flatten()[gen.sample_pixels(p.numel(), 256, meta["seed"])], z["pred_px"]) < 1e-6
    assert not bool(p.isnan().any()) and not bool(p.isinf().any())
