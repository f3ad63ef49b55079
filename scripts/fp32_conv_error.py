"""Forward rounding error of ONE 3x3x3 convolution against float64: this library's kernels (exact-fp32 MFMA, fp16x3, bf16x6)
and torch's fp32 CPU convolution (the reference path).  VERDICT r3 weak #1 asks why the exact-fp32 build sits further from
float64 than the reference's fp32 path on the depth-4 survey: the candidates are the statistics / norm fusions (switches, see
scripts/depth4_bisect.sh) and the accumulation ORDER of the convolutions -- v_mfma_f32_32x32x2_f32 is a sequential fmaf
chain over K = 27 Cin terms (error ~ sqrt(K) eps), oneDNN sums in 16-lane blocks (~ sqrt(K / 16) eps).
    python scripts/fp32_conv_error.py > profiles/r04_fp32_conv_error.txt"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from torch_em_amd import ops  # noqa: E402


def main():
    print("# relative L2 error of y = conv3d(xhat, w) against float64; xhat ~ N(0, 1) (a normalised activation), w ~ kaiming")
    print("# Cin Cout  voxels  K=27*Cin | cpu fp32 (reference path) | exact-fp32 MFMA | fp16x3 (default fwd) | bf16x6 | bf16x3")
    torch.manual_seed(0)
    for cin, cout, s in ((32, 32, 32), (64, 64, 24), (128, 128, 16), (256, 256, 16), (512, 512, 8)):
        x = torch.randn(1, cin, s, s, s)
        w = (torch.rand(cout, cin, 3, 3, 3) * 2 - 1) / (27 * cin) ** 0.5
        ref = F.conv3d(x.double(), w.double(), padding=1)
        e = lambda y: float((y.double() - ref).norm() / ref.norm())  # noqa: E731
        row = [e(F.conv3d(x, w, padding=1))]
        x5 = x.permute(0, 2, 3, 4, 1).contiguous().cuda()
        one = torch.ones(1, cin, device="cuda")
        zero = torch.zeros(1, cin, device="cuda")
        for mode in (1, 4, 3, 2):
            y5 = ops.new_act(1, s, s, s, cout, "cuda")
            # scale / shift = identity norm: mode 4 (fp16x3) is only defined for pre-normalised inputs
            ops.conv_fwd(x5, ops.pack_weights(w.cuda(), transpose=False, mfma=mode), None, y5, (3, 3, 3), cin, cout, scale=one,
                         shift=zero, mfma=mode)
            row.append(e(y5.permute(0, 4, 1, 2, 3).cpu()))
        print(f"{cin:4d} {cout:4d} {s ** 3:7d} {27 * cin:6d} | " + " | ".join(f"{v:.2e}" for v in row), flush=True)


if __name__ == "__main__":
    main()
