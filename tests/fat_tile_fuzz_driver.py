"""Run in a fresh interpreter with BSVD_FAT_MIN_WGS=1 (the Python executor reads it once and passes it as BsvdConvArgs.fat_min_wgs): every split-mode layer with more than 64
output channels then takes the 128-accumulator tile whatever its grid, so small random geometries reach its special paths -- waves
below the image for every H mod 16 in 1..8, zero-chunk skipping for T = 1 (both temporal neighbours missing), 2 and 3, every halo form,
the PixelShuffle epilogue, masked output columns.  Prints one line per case and 'FUZZ OK n'."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(HERE, "golden"))
assert os.environ.get("BSVD_FAT_MIN_WGS") == "1"
import test_gpu_f16x3 as S  # noqa: E402


def draw(rs):
    epi = int(rs.choice([0, 0, 1]))
    tsm = bool(epi == 0 and rs.rand() < 0.6)
    cin = int(rs.choice([128, 256]))
    cout = cin if tsm else int(rs.choice([128, 256]))
    act = str(rs.choice(["relu6", "relu", "none"])) if epi == 0 else "none"
    T = int(rs.randint(1, 4))
    H, W = int(rs.randint(1, 41)), int(rs.randint(1, 36))
    if cin * cout >= 256 * 256:
        H, W = min(H, 24), min(W, 20)
    return cin, cout, 1, tsm, act, epi, T, H, W


def nets(n):
    """whole bsvd_c64 / c32-sized networks on small random clips with every wide layer on the fat tile: clip vs the oracle in both
    precisions, stream schedule bit-identical (tests/test_gpu_fuzz.py::test_random_clip_whole_network under the override)"""
    import bsvd_amd.arch as A
    import test_gpu_fuzz as F
    A.WIDE_CONV_DEFAULT = "direct"      # the direct form of the wide layers (the fat tile) for every model this process builds
    for seed in range(n):
        print("net case", seed, flush=True)
        F.test_random_clip_whole_network(seed)
    print("FUZZ NETS OK", n)


def main():
    if len(sys.argv) > 2 and sys.argv[2] == "nets":
        return nets(int(sys.argv[1]))
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    import torch
    from bsvd_amd.engine import HipExecutor
    seen = set()
    orig = HipExecutor.conv

    def spy(self, sp, *a, **k):
        self.record_variants = True
        y = orig(self, sp, *a, **k)
        if sp.key == "l":
            seen.add(self.last_variant)
        return y

    HipExecutor.conv = spy
    for seed in range(n):
        args = draw(np.random.RandomState(7000 + seed))
        print("case", seed, args, flush=True)
        S.test_layer_split_vs_oracle(*args)
    print("variants", sorted(seen))
    assert seen == {"conv3x3_kernel<4,2,2,2,1>[f16x3]"}, seen
    torch.cuda.synchronize()
    print("FUZZ OK", n)


if __name__ == "__main__":
    main()
