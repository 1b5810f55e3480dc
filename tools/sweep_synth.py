#!/usr/bin/env python3
"""Developer tool: sweep the synthetic generator's parameters at the bench shape and print
rank / FPR@95 / loss after N steps, to pick a workload whose operating point resembles the
reference's logs (rank ~64-71 and FPR@95 5-10 % at step 500:
workspace/pj-learn/logging/liberty-liberty-0.035-0.250-pr#7-0.0010-0.100-pj.log:15-24)."""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from bench import make_U  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--F", type=int, default=8192)
    ap.add_argument("--N", type=int, default=500000)
    ap.add_argument("--steps", type=int, default=520)
    ap.add_argument("--cases", type=str, required=True,
                    help="semicolon-separated latent,mu,sigma_pos,noise,jitter tuples")
    args = ap.parse_args()
    dlco = importlib.import_module("opencv-dlco_amd")
    for case in args.cases.split(";"):
        latent, mu, sp, noise, jit = case.split(",")
        latent, mu, sp, noise, jit = int(latent), float(mu), float(sp), float(noise), float(jit)
        ctx = dlco.Context(args.F, args.N, B=200, mu=mu, gamma=0.5)
        ctx.synth_data(make_U(args.F, latent, 2215 + 1), 2215 + 1, sp, 1.0, noise, jit)
        t0 = time.perf_counter()
        ctx.steps(args.steps - 200)
        ctx.sync()
        t1 = time.perf_counter()
        ctx.steps(200)
        ctx.sync()
        t2 = time.perf_counter()
        e = ctx.log_step()
        cn = ctx.counters()
        print(json.dumps(dict(latent=latent, mu=mu, sigma_pos=sp, noise=noise, jitter=jit, rank=e.rank, fpr95=round(e.fpr95, 4),
                              auc=round(e.auc, 5), loss=round(e.loss_val, 5), regul=round(e.regul, 5),
                              ms_per_step_last200=round((t2 - t1) / 200 * 1e3, 3), burn_s=round(t1 - t0, 2),
                              nonconv=cn["nonconverged"], k_mean=round(cn["active_rows"] / max(cn["steps"], 1), 1))), flush=True)
        ctx.close()


if __name__ == "__main__":
    main()
