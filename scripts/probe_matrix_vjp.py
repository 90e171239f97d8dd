"""Pullbacks of VecCorrBijector / PDVecBijector, both directions, one dtype: CALL time (HIP events around bj.vjp, host side included;
median of 7 after 3 warm-up calls) and percent of 8 TB/s on the algorithmic bytes of scripts/bench_rows.py.  For same-box A/B of the switches
(BJX_MATRIX_VJP_MFMA=0/1/2, BJX_MATRIX_VJP_GRP=0):   python scripts/probe_matrix_vjp.py [--dtype float64] [--ks 12,16,24,32]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bijectors_amd as bj  # noqa: E402


def randn(r, n, dev, seed, dt, std=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (std * torch.randn(n, r, generator=g, dtype=torch.float64)).to(dt).to(dev).T


def timed(fn):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="float32")
    ap.add_argument("--ks", default="12,16,24,32,48,64")
    a = ap.parse_args()
    dt = getattr(torch, a.dtype)
    sz = 4 if dt == torch.float32 else 8
    dev = torch.device("cuda", 0)
    lb = {12: 19, 16: 18, 24: 16, 32: 16, 48: 14, 64: 14}
    print(f"dtype {a.dtype}  BJX_MATRIX_VJP_MFMA={os.environ.get('BJX_MATRIX_VJP_MFMA', '-')}  BJX_MATRIX_VJP_GRP={os.environ.get('BJX_MATRIX_VJP_GRP', '-')}")
    for K in [int(k) for k in a.ks.split(",")]:
        N = 1 << (lb.get(K, 16) - (0 if sz == 4 else 1))
        for nm, cls in (("VecCorr", bj.VecCorrBijector), ("PDVec", bj.PDVecBijector)):
            b = cls()
            nv = b._n(K)
            y = randn(nv, N, dev, 40, dt, std=0.3)
            X = bj.transform(bj.inverse(b), y)
            Xbar = randn(K * K, N, dev, 42, dt).T.reshape(N, K, K).permute(2, 1, 0)
            ybar = randn(nv, N, dev, 43, dt)
            lbar = randn(N, 1, dev, 44, dt).reshape(-1).contiguous()
            ms_i = timed(lambda: bj.vjp(bj.inverse(b), y, Xbar, lbar))
            ms_f = timed(lambda: bj.vjp(b, X, ybar, lbar))
            by_i = sz * (2 * nv + K * K) + sz
            by_f = sz * (2 * K * K + nv) + sz
            print(f"  {nm:8s} K={K:2d} N=2^{N.bit_length() - 1}: inverse {ms_i:.4f} ms {100 * by_i * N / (ms_i * 1e-3) / 8e12:5.1f} %   forward {ms_f:.4f} ms {100 * by_f * N / (ms_f * 1e-3) / 8e12:5.1f} %")


if __name__ == "__main__":
    main()
