"""GPU parity of the conditioning encoders and of the WHOLE reference chain
(scripts/inference_test_bench.py:441-495: landmarks -> conditioning_with_feat -> VAE encode + posterior sample ->
DDIM (CFG 3.5) -> VAE decode -> clamp) against outputs of the reference itself (tests/golden), through the
reference's own class / method surface built from a YAML registry config."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from reface_amd import params as P
from reface_amd.params import seeded_randn as rnd

pytestmark = pytest.mark.gpu
DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL_CLIP = dict(hidden=128, intermediate=512, layers=2, heads=4)


def G(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def maxerr(a, b):
    a = a.detach().float().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    return float(np.abs(a.astype(np.float64) - np.asarray(b).astype(np.float64)).max())


def test_arcface_vs_reference_golden(golden_dir):
    from reface_amd.encoders import Backbone
    g = G(golden_dir, "arcface")
    net = Backbone(input_size=112, num_layers=50, drop_ratio=0.6, mode="ir_se")
    net.load_state_dict(P.seeded_state_dict(P.arcface_param_specs(), 77), strict=True)
    net.to(DEV)
    f112 = net(rnd((2, 3, 112, 112), 51).to(DEV))[0]
    assert maxerr(f112, g["feats112"]) < 1e-5, maxerr(f112, g["feats112"])
    f = net.forward_from_clip_image(rnd((2, 3, 224, 224), 50).to(DEV))[0]
    assert maxerr(f, g["feats"]) < 1e-5, maxerr(f, g["feats"])
    assert torch.allclose(f.norm(dim=1).cpu(), torch.ones(2), atol=1e-6)


def test_clip_vs_reference_golden(golden_dir):
    from reface_amd.encoders import FrozenCLIPEmbedder
    g = G(golden_dir, "clip_small")
    m = FrozenCLIPEmbedder(vision_config=SMALL_CLIP)
    m.load_state_dict(P.seeded_state_dict(P.clip_param_specs(m.cfg), 88), strict=True)
    m.to(DEV)
    z = m.encode(rnd((2, 3, 224, 224), 60).to(DEV))
    assert z.shape == (2, 1, 768)
    assert maxerr(m._engine(2).pooled, g["pooled"]) < 5e-5
    assert maxerr(z, g["z"]) < 5e-5, maxerr(z, g["z"])
    g = G(golden_dir, "clip_l14_1layer")                      # full ViT-L/14 width (1024 / 16 heads / 257 tokens), 1 layer
    m = FrozenCLIPEmbedder(vision_config=dict(layers=1))
    m.load_state_dict(P.seeded_state_dict(P.clip_param_specs(m.cfg), 89), strict=True)
    m.to(DEV)
    z = m.encode(rnd((1, 3, 224, 224), 61).to(DEV))
    assert maxerr(z, g["z"]) < 5e-5, maxerr(z, g["z"])


def _small_pipeline():
    from ldm.util import instantiate_from_config
    from reface_amd import config as rcfg
    cfg = rcfg.load(os.path.join(ROOT, "tests", "configs", "reface_small.yaml"))
    cfg.model.params.cond_stage_config["params"] = {"vision_config": SMALL_CLIP}
    model = instantiate_from_config(cfg.model)
    sd = {}
    sd.update(P.seeded_state_dict(P.unet_param_specs(model.model.diffusion_model.cfg), 7, "model.diffusion_model."))
    sd.update(P.seeded_state_dict(P.vae_param_specs(model.first_stage_model.cfg), 55, "first_stage_model."))
    sd.update(P.seeded_state_dict(P.clip_param_specs(model.cond_stage_model.cfg), 88, "cond_stage_model."))
    # the fixture's ArcFace weights were seeded on the un-prefixed keys (they went through an arcface_path file)
    sd.update({"face_ID_model.facenet." + k: v for k, v in P.seeded_state_dict(P.arcface_param_specs(), 77).items()})
    sd.update(P.seeded_state_dict(P.cond_head_specs(), 9))
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected
    model.cuda()
    model.eval()
    return model


def test_whole_chain_vs_reference_golden(golden_dir):
    from ldm.models.diffusion.ddim import DDIMSampler
    from reface_amd import ops
    g = G(golden_dir, "e2e_small")
    model = _small_pipeline()
    B, H = 2, 256
    target = torch.tanh(rnd((B, 3, H, H), 70))
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(H), indexing="ij")
    ell = (((yy - H / 2) / (0.30 * H)) ** 2 + ((xx - H / 2) / (0.38 * H)) ** 2) <= 1.0
    inpaint_mask = (~ell).float()[None, None].repeat(B, 1, 1, 1)
    inpaint_image = target * inpaint_mask
    ref = rnd((B, 3, 224, 224), 71)
    x_T = rnd((B, 4, H // 8, H // 8), 72)

    sampler = DDIMSampler(model)
    uc = model.learnable_vector.repeat(B, 1, 1)
    assert maxerr(uc, g["uc"]) == 0
    landmarks = model.get_landmarks(target)                    # no dlib here -> the no-face branch, as in the fixture
    assert maxerr(landmarks, g["landmarks"]) < 1e-6
    c = model.conditioning_with_feat(ref.to(DEV), landmarks=landmarks, tar=target.to(DEV)).float()
    assert c.shape == (B, 1, 768)
    assert maxerr(c, g["c"]) < 5e-5, maxerr(c, g["c"])
    post = model.encode_first_stage(inpaint_image.to(DEV))
    assert maxerr(post.mean, g["post_mean"]) < 1e-4 and maxerr(post.logvar, g["post_logvar"]) < 1e-4
    z_inpaint = model.get_first_stage_encoding(post, noise=torch.from_numpy(g["eps"]))
    assert maxerr(z_inpaint, g["z_inpaint"]) < 5e-5
    m64 = torch.empty((B, 1, H // 8, H // 8), dtype=torch.float32, device=DEV)
    ops.bilinear_resize(inpaint_mask.to(DEV), m64)()
    assert maxerr(m64, g["mask64"]) == 0
    samples, _ = sampler.sample(S=5, conditioning=c, batch_size=B, shape=[4, H // 8, H // 8], verbose=False,
                                unconditional_guidance_scale=3.5, unconditional_conditioning=uc, eta=0.0, x_T=x_T.to(DEV),
                                test_model_kwargs={"inpaint_image": z_inpaint, "inpaint_mask": m64})
    assert maxerr(samples, g["samples"]) < 3e-4, maxerr(samples, g["samples"])
    x_dec = model.decode_first_stage(samples)
    e = maxerr(x_dec, g["x_dec"])
    assert e < 1e-3, e                                        # the north-star pixel bound (images are in [-1, 1] here)
    img = torch.empty_like(x_dec)
    ops.to_image(x_dec, img)()
    u8 = (255.0 * img.permute(0, 2, 3, 1).cpu().numpy()).astype(np.uint8)
    assert (np.abs(u8.astype(int) - g["u8"].astype(int)) <= 1).all() and (u8 != g["u8"]).mean() < 2e-3
    # the on-disk output tree (inference_test_bench.py:500-552) of item 0 against what the reference's own save block wrote
    from reface_amd import output as O
    gp = G(golden_dir, "e2e_png")
    ref_big = torch.empty((B, 3, H, H), dtype=torch.float32, device=DEV)
    ops.bilinear_resize(ref.to(DEV).contiguous(), ref_big)()
    o = O.compose(img[0].cpu().numpy(), target[0].numpy(), inpaint_image[0].numpy(), inpaint_mask[0].numpy(), ref_big[0].cpu().numpy())
    assert np.array_equal(o["mask"], gp["mask"])
    panel = lambda k: gp["grid"][2:2 + H, 2 + k * (H + 2):2 + k * (H + 2) + H]
    assert np.array_equal(o["GT"], panel(0)) and np.array_equal(o["inpaint"], panel(1))
    assert np.array_equal(o["ref"], panel(2)), np.abs(o["ref"].astype(int) - panel(2).astype(int)).max()      # rf_bilinear_resize is exact
    assert (np.abs(o["result"].astype(int) - panel(3).astype(int)) <= 1).all()
    d = np.abs(o["grid"].astype(int) - gp["grid"].astype(int))
    assert d.max() <= 1 and (d[:, :2 + 3 * (H + 2)] == 0).all()                # everything left of the result panel is byte-exact


SMALL_UNET = dict(in_channels=9, model_channels=64, out_channels=4, num_res_blocks=2, attention_resolutions=(4, 2, 1), channel_mult=(1, 2, 4, 4),
                  num_heads=8, context_dim=768)
SMALL_VAE = dict(ch=32, ch_mult=(1, 2, 4, 4), num_res_blocks=2, in_channels=3, out_ch=3, z_channels=4, embed_dim=4, double_z=True,
                 attn_resolutions=(), resolution=256)


def _oracle_chain_check(dump_npz, png_path, S, scale, rows=(0,)):
    """The CPU oracle chain (oracle/: conditioning_with_feat -> VAE encode + posterior sample -> mask64 -> CFG DDIM -> fp32 decode ->
    uint8) on exactly the tensors a caller's batch fed the engines (SwapRunner --dump_tensors), with the CLI's `--ckpt none` seeds;
    checks the engines' intermediates and the PNG the caller wrote (<= 1 LSB)."""
    from PIL import Image
    from oracle import ddim as oddim, encoders as oenc, unet as ounet, vae as ovae
    d = np.load(dump_npz)
    ucfg, vcfg = P.UNetConfig(**SMALL_UNET), P.VAEConfig(**SMALL_VAE)
    ccfg = P.CLIPVisionConfig(**dict(SMALL_CLIP, patch=14, image=224, proj=768, mapper_layers=5))
    # (seeded tensors depend on the FULL key: generate with the prefixes load_model_from_config(--ckpt none) uses, then strip them)
    strip = lambda sd, pre: {k[len(pre):]: v for k, v in sd.items()}
    usd = strip(P.seeded_state_dict(P.unet_param_specs(ucfg), 1234, "model.diffusion_model."), "model.diffusion_model.")
    vsd = strip(P.seeded_state_dict(P.vae_param_specs(vcfg), 55, "first_stage_model."), "first_stage_model.")
    csd = strip(P.seeded_state_dict(P.clip_param_specs(ccfg), 88, "cond_stage_model."), "cond_stage_model.")
    asd = strip(P.seeded_state_dict(P.arcface_param_specs(), 77, "face_ID_model.facenet."), "face_ID_model.facenet.")
    heads = P.seeded_state_dict(P.cond_head_specs(), 9)
    r = list(rows)
    T = lambda k: torch.from_numpy(d[k][r])
    target, ref, inp_img, inp_mask, x_T, noise = T("test_batch"), T("ref_imgs"), T("inpaint_image"), T("inpaint_mask"), T("x_T"), T("post_noise")
    B = len(r)
    torch.set_num_threads(max(1, min(32, len(os.sched_getaffinity(0)))))
    with torch.no_grad():
        c = oenc.conditioning_with_feat(heads, csd, ccfg, asd, P.arcface_units(), ref, torch.zeros(B, 136), target)      # no dlib: zeros branch
        assert maxerr(c, d["c"][r]) < 5e-5, maxerr(c, d["c"][r])
        mean, logvar = ovae.encode_moments(vsd, vcfg, inp_img)
        z_inp = ovae.first_stage_encoding(mean, logvar, noise)
        assert maxerr(z_inp, d["z_inpaint"][r]) < 1e-4, maxerr(z_inp, d["z_inpaint"][r])
        m64 = oenc.mask64(inp_mask)
        assert maxerr(m64, d["mask64"][r]) == 0
        uc = heads["learnable_vector"].repeat(B, 1, 1)
        plan = ounet.plan_of(usd, ucfg.num_heads)
        samples, _ = oddim.sample(lambda x, t, cc: ounet.unet_forward(usd, plan, x, t, cc, ucfg.model_channels), S, x_T, c, uc, z_inp, m64, scale)
        assert maxerr(samples, d["samples"][r]) < 5e-4, maxerr(samples, d["samples"][r])
        x_dec = ovae.decode_first_stage(vsd, vcfg, samples)
    img01 = torch.clamp((x_dec + 1.0) / 2.0, 0.0, 1.0)
    assert maxerr(img01, d["x_img"][r]) < 1e-3, maxerr(img01, d["x_img"][r])          # the north-star pixel bound
    u8 = oenc.to_uint8_image(x_dec)[0]
    png = np.asarray(Image.open(png_path))
    if png.shape != u8.shape:           # the video caller stores its crops at 1024^2 (PIL bilinear of the 512^2 result): compare at 512^2
        return u8
    diff = np.abs(png.astype(int) - u8.astype(int))
    assert diff.max() <= 1 and (diff != 0).mean() < 5e-3, (diff.max(), (diff != 0).mean())
    return u8


def test_cli_synthetic_run(tmp_path):
    """scripts/inference_test_bench.py end to end on seeded weights: writes the reference's output tree."""
    import json
    out = tmp_path / "out"
    cmd = [sys.executable, os.path.join(ROOT, "scripts", "inference_test_bench.py"), "--outdir", str(out), "--config",
           os.path.join(ROOT, "tests", "configs", "reface_small.yaml"), "--ckpt", "none", "--dataset", "synthetic", "--n_items", "3",
           "--n_samples", "2", "--ddim_steps", "5", "--scale", "3.5", "--H", "256", "--W", "256", "--clip_vision_config", json.dumps(SMALL_CLIP)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    res = sorted(os.listdir(out / "results"))
    assert res == ["000000000000.png", "000000000001.png", "000000000002.png"]
    assert len(os.listdir(out / "grid")) == 3
    assert sorted(os.listdir(out / "samples")) == sorted(f"{i:012d}_{k}.png" for i in range(3) for k in ("mask", "GT", "inpaint", "ref"))
    from PIL import Image
    im = np.asarray(Image.open(out / "results" / res[0]))
    assert im.shape == (256, 256, 3) and im.std() > 1.0
    grid = np.asarray(Image.open(out / "grid" / ("grid-" + res[0])))
    assert grid.shape == (260, 4 * 258 + 2, 3)                                  # make_grid of 4 panels, padding 2
    for k, nm in enumerate(("_GT", "_inpaint", "_ref")):
        pn = np.asarray(Image.open(out / "samples" / (res[0][:-4] + nm + ".png")))
        assert np.array_equal(grid[2:258, 2 + k * 258:2 + k * 258 + 256], pn), nm
    assert np.array_equal(grid[2:258, 2 + 3 * 258:2 + 3 * 258 + 256], im)
    mk = np.asarray(Image.open(out / "samples" / (res[0][:-4] + "_mask.png")))
    assert set(np.unique(mk)) == {127, 255}                                     # 255 * (mask + 1) / 2


def test_cli_plms_start_from_target(tmp_path):
    """The 'next' rows 8f.4: --plms and --Start_from_target through the same CLI, in throughput mode (--precision bf16: bf16 UNet,
    CLIP / ArcFace towers and VAE encoder; fp32 VAE decode)."""
    import json
    out = tmp_path / "out"
    cmd = [sys.executable, os.path.join(ROOT, "scripts", "inference_test_bench.py"), "--outdir", str(out), "--config",
           os.path.join(ROOT, "tests", "configs", "reface_small.yaml"), "--ckpt", "none", "--dataset", "synthetic", "--n_items", "2",
           "--n_samples", "2", "--ddim_steps", "5", "--scale", "3.5", "--H", "256", "--W", "256", "--plms", "--Start_from_target", "--precision", "bf16",
           "--target_start_noise_t", "800", "--clip_vision_config", json.dumps(SMALL_CLIP)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert sorted(os.listdir(out / "results")) == ["000000000000.png", "000000000001.png"]


def test_cli_precision_fp16_close_to_full(tmp_path):
    """--precision fp16 (round 6): the drop-in CLI with the UNet on fp16 operands (fp32 towers / VAE encoder, split-bf16 decode) against --precision full on
    the same synthetic pairs: the saved results/<id>.png agree to a couple of grey levels (the bf16 mode is an order of magnitude further away)."""
    import json
    import numpy as np
    from PIL import Image
    imgs = {}
    for prec in ("full", "fp16"):
        out = tmp_path / prec
        cmd = [sys.executable, os.path.join(ROOT, "scripts", "inference_test_bench.py"), "--outdir", str(out), "--config",
               os.path.join(ROOT, "tests", "configs", "reface_small.yaml"), "--ckpt", "none", "--dataset", "synthetic", "--n_items", "2",
               "--n_samples", "2", "--ddim_steps", "5", "--scale", "3.5", "--H", "256", "--W", "256", "--precision", prec,
               "--clip_vision_config", json.dumps(SMALL_CLIP)]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
        imgs[prec] = [np.asarray(Image.open(out / "results" / f)).astype(np.int32) for f in sorted(os.listdir(out / "results"))]
    assert len(imgs["full"]) == len(imgs["fp16"]) == 2
    d = max(int(np.abs(a - b).max()) for a, b in zip(imgs["full"], imgs["fp16"]))
    print(f"--precision fp16 vs full, saved PNGs: max |d| = {d} grey levels")
    assert d <= 3, d


def test_cli_swap_selected(tmp_path):
    """the selected-swap caller (inference_swap_selected.py:516-762) on a prepared <Base_dir> tree: every source onto every
    target, the reference's per-source output folders."""
    import json
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_host_cpu import _prepared_swap_tree
    base, out = str(tmp_path / "base"), tmp_path / "out"
    _prepared_swap_tree(base, n_tar=3, n_src=2)
    cmd = [sys.executable, os.path.join(ROOT, "scripts", "inference_swap_selected.py"), "--outdir", str(out), "--Base_dir", base, "--config",
           os.path.join(ROOT, "tests", "configs", "reface_small.yaml"), "--ckpt", "none", "--n_samples", "2", "--ddim_steps", "4", "--scale", "3.5",
           "--H", "512", "--W", "512", "--precision", "full", "--num_workers", "0", "--clip_vision_config", json.dumps(SMALL_CLIP),
           "--dump_tensors", str(tmp_path / "dump")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    for s in ("0", "1"):
        assert sorted(os.listdir(out / "results" / s)) == [f"{i:012d}.png" for i in range(3)]
        assert len(os.listdir(out / "grid" / s)) == 3 and len(os.listdir(out / s)) == 12
    assert any(f.startswith("_intermediate_") for f in os.listdir(out / "model_outputs"))
    # caller-specific arithmetic against the oracle: batches are (source 0: targets 0-1, target 2), (source 1: targets 0-1, target 2);
    # the LAST batch is source 1 on target 2 -- ONE source repeated over the batch (inference_swap_selected.py:649-653), its own folder
    dumps = sorted(os.listdir(tmp_path / "dump"))
    assert dumps == [f"batch_{i:04d}.npz" for i in range(4)], dumps
    d0, d2 = np.load(tmp_path / "dump" / dumps[0]), np.load(tmp_path / "dump" / dumps[2])
    assert np.array_equal(d0["ref_imgs"][0], d0["ref_imgs"][1]) and not np.array_equal(d0["ref_imgs"][0], d2["ref_imgs"][0])     # one source per pass
    assert np.array_equal(d0["test_batch"], d2["test_batch"])                                                                       # every source sees every target
    _oracle_chain_check(tmp_path / "dump" / dumps[3], out / "results" / "1" / "000000000002.png", S=4, scale=3.5)
    _oracle_chain_check(tmp_path / "dump" / dumps[0], out / "results" / "0" / "000000000000.png", S=4, scale=3.5)
    # without the prepared tree the CLI says what is out of scope instead of failing somewhere inside
    r2 = subprocess.run(cmd[:4] + ["--Base_dir", str(tmp_path / "nothing")] + cmd[6:], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r2.returncode != 0 and "stage 1" in (r2.stderr + r2.stdout)


def test_cli_swap_video(tmp_path):
    """the video caller's sampling stage (inference_swap_video.py:504-700) on the tree its stage 1 leaves: one source on every
    frame crop, whole batches only (drop_last), swapped crops at 1024^2 in model_outputs/."""
    import json
    import shutil
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_host_cpu import _prepared_swap_tree
    base, out = tmp_path / "base", tmp_path / "out"
    _prepared_swap_tree(str(base), n_tar=5, n_src=1)
    os.rename(base / "target_cropped", base / "clipcropped_face")
    os.rename(base / "mask_frames", base / "clipmask_frames")
    (out / "temp_results").mkdir(parents=True)
    shutil.copy(base / "source_cropped" / "0.png", out / "temp_results" / "me.png")
    shutil.copy(base / "source_mask" / "0.png", out / "temp_results" / "me.jpg")          # (the reference saves the label map under the source's own file name)
    cmd = [sys.executable, os.path.join(ROOT, "scripts", "inference_swap_video.py"), "--outdir", str(out), "--Base_dir", str(base), "--target_video",
           "videos/clip.mp4", "--src_image", "faces/me.jpg", "--config", os.path.join(ROOT, "tests", "configs", "reface_small.yaml"), "--ckpt", "none",
           "--n_samples", "2", "--ddim_steps", "4", "--scale", "3.5", "--precision", "full", "--num_workers", "0", "--clip_vision_config",
           json.dumps(SMALL_CLIP), "--dump_tensors", str(tmp_path / "dump")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert sorted(os.listdir(out / "model_outputs")) == [f"{i:012d}.png" for i in range(4)]          # 5 frames, batches of 2, drop_last
    from PIL import Image
    im = np.asarray(Image.open(out / "model_outputs" / "000000000003.png"))
    assert im.shape == (1024, 1024, 3)
    # this caller's quirks against the oracle chain: ONE start latent for every frame of every batch (inference_swap_video.py), whole
    # batches only; the 1024^2 crop is the PIL-bilinear enlargement of the 512^2 result the oracle reproduces
    dumps = sorted(os.listdir(tmp_path / "dump"))
    assert len(dumps) == 2
    d0, d1 = np.load(tmp_path / "dump" / dumps[0]), np.load(tmp_path / "dump" / dumps[1])
    assert np.array_equal(d0["x_T"][0], d0["x_T"][1]) and np.array_equal(d0["x_T"], d1["x_T"])
    u8 = _oracle_chain_check(tmp_path / "dump" / dumps[1], out / "model_outputs" / "000000000003.png", S=4, scale=3.5, rows=(1,))
    big = np.asarray(Image.fromarray(u8).resize((1024, 1024), Image.BILINEAR))
    diff = np.abs(big.astype(int) - im.astype(int))
    assert diff.max() <= 2 and (diff != 0).mean() < 2e-2, (diff.max(), (diff != 0).mean())
    r2 = subprocess.run(cmd[:6] + ["--target_video", "videos/other.mp4"] + cmd[8:], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r2.returncode != 0 and "stage 1" in (r2.stderr + r2.stdout)


def test_device_prep_matches_host(tmp_path):
    """SURVEY 8f.1 (second half): the dataset's tensors built on the GPU from uint8 arrays (reface_amd/prep.py, rf_u8_to_norm /
    rf_label_mask / rf_mul_mask / rf_bilinear_resize) are bit-identical to the host path of the same reader (target, keep-mask,
    masked target) resp. within one fp32 ulp (the masked source face, whose mask is bilinearly resized by a non-integer ratio)."""
    from PIL import Image
    from reface_amd.data import CelebAdataset
    from reface_amd.prep import DevicePrep
    root = tmp_path / "CelebAMask-HQ"
    (root / "CelebA-HQ-img").mkdir(parents=True)
    (root / "CelebA-HQ-mask" / "Overall_mask").mkdir(parents=True)
    rng = np.random.default_rng(0)
    for i in (28000, 28001, 29000, 29001):
        Image.fromarray(rng.integers(0, 256, (96, 80, 3), dtype=np.uint8)).save(root / "CelebA-HQ-img" / f"{i}.jpg")
        Image.fromarray(rng.integers(0, 19, (512, 512), dtype=np.uint8)).save(root / "CelebA-HQ-mask" / "Overall_mask" / f"{i}.png")
    for gray in (True, False):
        kw = dict(dataset_dir=str(root), n_targets=2, gray_outer_mask=gray)
        host, raw = CelebAdataset(**kw), CelebAdataset(raw=True, **kw)
        prep = DevicePrep(raw.remove_tar, raw.preserve_src, raw.gray_outer_mask)
        items = [raw[i] for i in range(2)]
        target, out = prep(*(torch.stack([it[k] for it in items]) for k in range(4)))
        torch.cuda.synchronize()
        # raw="full" (what --gpu_prep asks for): the source face arrives undecimated and rf_resize_u8_linear resizes it -- same tensors
        from reface_amd.data import raw_collate
        full = CelebAdataset(raw="full", **kw)
        fitems = raw_collate([full[i] for i in range(2)])
        assert tuple(fitems[2].shape[1:3]) == (96, 80)
        target_f, out_f = prep(*fitems[:4])
        torch.cuda.synchronize()
        assert torch.equal(target_f, target) and torch.equal(out_f["ref_imgs"], out["ref_imgs"]) and torch.equal(out_f["inpaint_image"], out["inpaint_image"])
        for i in range(2):
            t, _, hk, sid = host[i]
            assert sid == items[i][4]
            assert torch.equal(target[i].cpu(), t)
            assert torch.equal(out["inpaint_mask"][i].cpu(), hk["inpaint_mask"])
            assert torch.equal(out["inpaint_image"][i].cpu(), hk["inpaint_image"])
            # the source face goes through a 512 -> 224 bilinear resize of its mask (non-integer ratio): last-bit differences of the
            # interpolation weights against torch's CPU kernel, <= 1 fp32 ulp of the O(1) products
            assert (out["ref_imgs"][i].cpu() - hk["ref_imgs"]).abs().max().item() < 2e-6


def test_compose_outputs_u8_matches_host_composition():
    """rf_compose_outputs_u8 (the CLI's output panels + 4-panel grid as packed uint8 records, composed on the device) is bit-identical to
    reface_amd/output.compose -- the host restatement of the reference's save block (inference_test_bench.py:500-552), itself pinned to the
    reference's own output for the e2e fixture -- including the un-clamped reference panel, whose out-of-range values wrap in the cast."""
    from reface_amd import ops
    from reface_amd import output as O
    rng = np.random.default_rng(5)
    for (B, H, W, grid) in ((2, 64, 64, True), (3, 40, 56, True), (1, 512, 512, True), (2, 64, 64, False)):
        res = rng.random((B, 3, H, W), dtype=np.float32)
        tgt = np.tanh(rng.standard_normal((B, 3, H, W))).astype(np.float32)
        msk = (rng.random((B, 1, H, W)) > 0.5).astype(np.float32)
        ref = (rng.standard_normal((B, 3, H, W)) * 2.5).astype(np.float32)
        nbytes, lay = O.record_layout(H, W, with_grid=grid)
        rec = torch.zeros((B, nbytes), dtype=torch.uint8, device="cuda")
        f = lambda a: torch.from_numpy(a).cuda()
        ops.compose_outputs_u8(f(res), f(tgt), f(tgt * msk), f(msk), f(ref), rec, with_grid=grid)()
        torch.cuda.synchronize()
        got = rec.cpu().numpy()
        for i in range(B):
            o = O.compose(res[i], tgt[i], tgt[i] * msk[i], msk[i], ref[i], skip_grid=not grid)
            for k, (off, shp) in lay.items():
                assert np.array_equal(got[i, off:off + int(np.prod(shp))].reshape(shp), o[k]), (B, H, W, k)


def test_resize_u8_linear_kernel_matches_host():
    """rf_resize_u8_linear (cv2 INTER_LINEAR arithmetic on the GPU, --gpu_prep) is bit-identical to the host restatement
    `reface_amd.data.resize_u8_linear` -- down- and upscales, non-square, the exact 2:1 fast-area case, a strided batch -- and the
    `raw="full"` reader + DevicePrep (stacked and ragged batches) gives the tensors of the `raw=True` reader, which resizes on the host."""
    from reface_amd import ops
    from reface_amd.data import resize_u8_linear
    from reface_amd.prep import DevicePrep
    rng = np.random.default_rng(11)
    for (B, H, W, oh, ow) in ((2, 1024, 1024, 224, 224), (3, 300, 517, 224, 224), (1, 96, 80, 224, 224), (2, 448, 448, 224, 224), (1, 37, 53, 37, 53),
                              (2, 512, 512, 224, 224)):
        x = rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)
        xd = torch.from_numpy(x).cuda()
        out = torch.empty((B, oh, ow, 3), dtype=torch.uint8, device="cuda")
        ops.resize_u8_linear(xd, out)()
        torch.cuda.synchronize()
        for b in range(B):
            assert np.array_equal(out[b].cpu().numpy(), resize_u8_linear(x[b], oh, ow)), (B, H, W, oh, ow, b)
    prep = DevicePrep([1, 2, 4], [1, 2, 4, 13], True)
    refs = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for (h, w) in ((300, 200), (1024, 1024))]
    got = prep.resize_sources([torch.from_numpy(r) for r in refs])
    for i, r in enumerate(refs):
        assert np.array_equal(got[i].cpu().numpy(), resize_u8_linear(r, 224, 224))
    st = np.stack([refs[1], refs[1][::-1].copy()])
    got = prep.resize_sources(torch.from_numpy(st))
    assert np.array_equal(got[1].cpu().numpy(), resize_u8_linear(st[1], 224, 224))


def test_bench_two_ranks_share_gpu():
    """The N > 1 path of bench.py end to end on one GPU: `--gpus 2 --share-gpu` spawns torch.distributed.run as a child (gloo, both ranks
    on cuda:0), rank 0 generates the weights and broadcasts them flat, the ranks shard the pairs, time = max over ranks.  Checks the
    line the driver parses: n_gpus, a finite whole-job value, and that rank 1 ended up with rank 0's weights."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline", "--no-roofline", "--no-parity", "--no-conditioning", "--ddim-steps", "10", "--batch", "2"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 4 and d["scaling"] == "weak"
    assert d["value"] > 0 and d["value"] == d["value"] and d["ms_per_step"] > 0
    assert d["weights_identical_on_all_ranks"] is True and len(d["weights_checksum_per_rank"]) == 2, d.get("weights_checksum_per_rank")


def test_one_inference_caller(tmp_path):
    """SURVEY 8f.3, last caller: scripts/one_inference.py -- `process_images` / `run_inference` of the reference's one-pair endpoint
    (one_inference.py:447-487, 521-790) for aligned crops: the endpoint's file flow, stage 2 through the shared batch body, JPEG bytes out;
    the PNG behind them against the CPU oracle chain (<= 1 LSB)."""
    import json
    from PIL import Image
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    from test_host_cpu import _prepared_swap_tree
    src_tree = str(tmp_path / "prep")
    _prepared_swap_tree(src_tree, n_tar=1, n_src=1)
    import one_inference as OI
    base, out = str(tmp_path / "base"), str(tmp_path / "out")
    OI.configure(["--outdir", out, "--Base_dir", base, "--config", os.path.join(ROOT, "tests", "configs", "reface_small.yaml"), "--ckpt", "none",
                  "--n_samples", "1", "--H", "512", "--W", "512", "--precision", "full", "--num_workers", "0", "--clip_vision_config",
                  json.dumps(SMALL_CLIP), "--dump_tensors", str(tmp_path / "dump")])
    buf = OI.process_images(os.path.join(src_tree, "source_cropped", "0.png"), os.path.join(src_tree, "source_mask", "0.png"),
                            os.path.join(src_tree, "target_cropped", "0.png"), os.path.join(src_tree, "mask_frames", "0.png"), steps=4, scale=3.5)
    jpg = np.asarray(Image.open(buf))
    assert jpg.shape == (512, 512, 3)
    png = os.path.join(out, "results", "0", "000000000000.png")
    _oracle_chain_check(tmp_path / "dump" / "batch_0000.npz", png, S=4, scale=3.5)
    import io
    again = io.BytesIO()
    Image.open(png).save(again, "JPEG")                                    # the endpoint returns exactly this encoding of the result PNG
    again.seek(0)
    assert np.array_equal(jpg, np.asarray(Image.open(again)))
