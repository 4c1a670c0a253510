"""Evaluation harness: PSNR pinned to the reference's own metric functions, tensor2img rounding,
SSIM against an independent direct implementation, ValFolderDataset reading/ordering/noise semantics."""
import os

import numpy as np
import pytest
import torch

from helpers import load_golden
from bsvd_amd import evaluation as E


def test_psnr_matches_reference_metrics():
    g = load_golden("g9_psnr")
    gt, out = torch.from_numpy(g["gt"]), torch.from_numpy(g["out"])
    assert abs(E.calculate_psnr(E.tensor2img(out), E.tensor2img(gt), crop_border=2) - float(g["psnr_u8"])) < 1e-9
    assert abs(E.calculate_psnr(E.tensor2img(out), E.tensor2img(gt), crop_border=0) - float(g["psnr_u8_nocrop"])) < 1e-9
    assert abs(E.calculate_psnr_float(out, gt, crop_border=2) - float(g["psnr_float"])) < 1e-5
    assert E.calculate_psnr(E.tensor2img(gt), E.tensor2img(gt), 2) == float("inf")


def test_tensor2img_clamps_rounds_and_swaps_channels():
    t = torch.tensor([[[0.5 / 255, 1.5 / 255]], [[-0.2, 0.4999 / 255]], [[2.0, 254.5 / 255]]])   # [3,1,2]
    img = E.tensor2img(t)
    assert img.dtype == np.uint8 and img.shape == (1, 2, 3)
    # numpy rounds half to even: 0.5 -> 0, 1.5 -> 2, 254.5 -> 254; BGR order
    assert img[0, 0].tolist() == [255, 0, 0] and img[0, 1].tolist() == [254, 0, 2]


def test_ssim_against_the_reference_golden():
    """g12: the reference's own _ssim / calculate_ssim (psnr_ssim.py:49-128) on seeded uint8 pairs."""
    g = load_golden("g12_ssim")
    for tag in "abc":
        gt, img = g["gt_" + tag], g["img_" + tag]
        assert abs(E.calculate_ssim(img, gt, crop_border=2) - float(g["ssim_crop2_" + tag])) < 1e-10
        assert abs(E.calculate_ssim(img, gt, crop_border=0) - float(g["ssim_crop0_" + tag])) < 1e-10
        chw = E.calculate_ssim(img.transpose(2, 0, 1), gt.transpose(2, 0, 1), crop_border=2, input_order="CHW")
        assert abs(chw - float(g["ssim_chw_" + tag])) < 1e-10


def test_ssim_against_direct_window_sum():
    rs = np.random.RandomState(3)
    a = rs.randint(0, 256, (20, 24, 3)).astype(np.uint8)
    b = np.clip(a.astype(np.int32) + rs.randint(-20, 21, a.shape), 0, 255).astype(np.uint8)
    k = E._gauss_kernel()
    win = np.outer(k, k)

    def direct(x, y):
        x, y = x.astype(np.float64), y.astype(np.float64)
        c1, c2 = 6.5025, 58.5225
        vals = []
        for i in range(x.shape[0] - 10):
            for j in range(x.shape[1] - 10):
                px, py = x[i:i + 11, j:j + 11], y[i:i + 11, j:j + 11]
                m1, m2 = (win * px).sum(), (win * py).sum()
                s1, s2 = (win * px * px).sum() - m1 * m1, (win * py * py).sum() - m2 * m2
                s12 = (win * px * py).sum() - m1 * m2
                vals.append((2 * m1 * m2 + c1) * (2 * s12 + c2) / ((m1 * m1 + m2 * m2 + c1) * (s1 + s2 + c2)))
        return np.mean(vals)

    want = np.mean([direct(a[2:-2, 2:-2, c], b[2:-2, 2:-2, c]) for c in range(3)])
    assert abs(E.calculate_ssim(a, b, crop_border=2) - want) < 1e-9
    assert abs(E.calculate_ssim(a, a, crop_border=0) - 1.0) < 1e-12


def test_val_folder_dataset(tmp_path):
    from PIL import Image
    rs = np.random.RandomState(1)
    for clip, n in (("b_clip", 3), ("a_clip", 12)):
        os.makedirs(tmp_path / clip)
        for i in range(n):
            Image.fromarray(rs.randint(0, 256, (8, 12, 3)).astype(np.uint8)).save(tmp_path / clip / ("%d.png" % i))
    opt = {"valsetdir": str(tmp_path), "num_validation_frames": 10, "valnoisestd": 30, "name": "t"}
    ds = E.ValFolderDataset(opt, device=torch.device("cpu"))
    assert ds.base_folder == ["a_clip", "b_clip"] and ds.num_frames == [10, 3]
    names = [os.path.basename(p) for p in E.image_names(str(tmp_path / "a_clip"))]
    assert names[:12] == ["%d.png" % i for i in range(12)]           # numeric, not lexicographic (10 after 9)
    torch.manual_seed(10)
    item = ds[0]
    assert item["gt"].shape == (1, 10, 3, 8, 12) and item["lq"].shape == item["gt"].shape
    assert item["noise_map"].shape == (1, 10, 1, 8, 12) and abs(float(item["noise_map"][0, 0, 0, 0, 0]) - 30 / 255) < 1e-7
    torch.manual_seed(10)
    want_noise = torch.FloatTensor(item["gt"].size()).normal_(mean=0, std=30 / 255.0)
    assert torch.equal(item["lq"], item["gt"] + want_noise)
    with Image.open(tmp_path / "a_clip" / "3.png") as im:
        assert np.array_equal((item["gt"][0, 3].numpy() * 255).round().astype(np.uint8), np.asarray(im).transpose(2, 0, 1))
    assert "noise_map" not in E.ValFolderDataset(dict(opt, blind=True), device=torch.device("cpu"))[1]


def test_seeded_construction_matches_reference_rng_stream():
    """Same weights and same generator state after construction as the reference classes (golden g11): a seeded
    evaluation therefore adds the reference's noise realisation."""
    from collections import OrderedDict
    from seeded import state_digest
    import bsvd_amd
    from bsvd_amd.arch import TSN
    g = load_golden("g11_seeded_init")
    torch.manual_seed(123)
    m = bsvd_amd.BSVD(precision="fp32", chns=[64, 128, 256], mid_ch=64, norm="none", act="relu6", interm_ch=64, pretrain_ckpt=None)
    assert state_digest(OrderedDict((k, v.numpy()) for k, v in m.state_dict().items())) == str(g["bsvd_digest"])
    assert np.array_equal(torch.rand(4).numpy(), g["bsvd_next"])
    torch.manual_seed(321)
    t = TSN(precision="fp32", num_segments=11, net2d_opt=dict(chns=[64, 128, 256], mid_ch=64, norm="none", act="relu", interm_ch=30, blind=True))
    assert state_digest(OrderedDict((k, v.numpy()) for k, v in t.state_dict().items())) == str(g["tsn_digest"])
    assert np.array_equal(torch.rand(4).numpy(), g["tsn_next"])
