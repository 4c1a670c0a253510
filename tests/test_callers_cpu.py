"""Caller harness against the golden captured from the reference's own
DenoisingModel.padding_input / crop_output and denoise_seq / temp_denoise (tests/golden/g8_pad_crop_clamp.npz)."""
import numpy as np
import pytest
import torch

from helpers import load_golden
from bsvd_amd.denoise import crop_padding, denoise_seq, pad_to_multiple_of_4, temp_denoise


class Dummy(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.p = torch.nn.Parameter(torch.zeros(1))
        self.seen = None

    def forward(self, x, noise_map=None):
        self.seen = (tuple(x.shape), None if noise_map is None else tuple(noise_map.shape),
                     None if noise_map is None else float(noise_map.flatten()[0]))
        return x * 3.0 - 1.0


def test_pad_and_crop_match_reference():
    g = load_golden("g8_pad_crop_clamp")
    lq = torch.from_numpy(g["lq"])
    padded, plist = pad_to_multiple_of_4(lq)
    assert plist == list(g["padding_list"])
    assert torch.equal(padded, torch.from_numpy(g["padded"]))
    assert torch.equal(padded[:, :, 30], padded[:, :, 28]) and torch.equal(padded[..., 51], padded[..., 47])
    assert torch.equal(crop_padding(padded[None], plist), torch.from_numpy(g["cropped"]))
    same, pl = pad_to_multiple_of_4(torch.zeros(2, 3, 8, 12))
    assert same.shape == (2, 3, 8, 12) and pl == [0] * 6


def test_denoise_seq_single_call_constant_sigma_clamp():
    g = load_golden("g8_pad_crop_clamp")
    seq = torch.from_numpy(g["seq"])
    nm = torch.full((7, 1, 8, 8), 30.0 / 255.0)
    m = Dummy()
    den = denoise_seq(seq, nm, -1, m)
    assert torch.equal(den, torch.from_numpy(g["den"]))
    assert list(m.seen[0]) == list(g["seen_x"]) and list(m.seen[1]) == list(g["seen_nm"])
    assert abs(m.seen[2] - float(g["seen_sigma"])) < 1e-8
    assert float(den.min()) >= 0.0 and float(den.max()) <= 1.0
    with pytest.raises(AssertionError):
        bad = nm.clone()
        bad[3] = 0.5
        temp_denoise(m, seq, bad)


def test_denoise_seq_segments_with_mirrored_tail():
    rs = np.random.RandomState(5)
    seq = torch.from_numpy(rs.uniform(0, 1, (7, 3, 4, 4)).astype(np.float32))
    calls = []

    class Rec(Dummy):
        def forward(self, x, noise_map=None):
            calls.append(x[0, :, 0, 0, 0].clone())
            return x

    den = denoise_seq(seq, None, 3, Rec())
    assert torch.equal(den, seq)
    assert [len(c) for c in calls] == [3, 3, 3]
    # tail = last frame + mirror of the two frames before it (validation_seq_infer.py:77-78)
    assert torch.equal(calls[2], torch.stack([seq[6, 0, 0, 0], seq[5, 0, 0, 0], seq[4, 0, 0, 0]]))
