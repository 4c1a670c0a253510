#!/usr/bin/env python3
"""Counterpart of the reference's ``python run_test.py -opt options/test/bsvd_c64.yml``
(/root/reference/run_test.py -> Experimental_root/scripts/test.py:12-16 -> BasicSR/basicsr/test.py:11-40) for the MI355X
engine: reads the SAME YAML (datasets of type ValFolderDataset, network_g, path, val.metrics), seeds like BasicSR
(manual_seed), iterates the datasets in sorted key order (test.py:27 -- the noise realisation depends on it), runs
DenoisingModel.test on every clip and prints per-folder / mean PSNR, PSNR-float, SSIM.

Data and checkpoints are not in the reference tree (dangling symlinks); point ``valsetdir`` / ``pretrain_ckpt`` at real
files.  ``--precision f16x3`` selects the split-fp16 mode; ``--force_yml network_g:engine_mode=stream`` style
overrides work like BasicSR's."""
import argparse
import json
import os
import random
import sys

import numpy as np
import torch
import yaml

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bsvd_amd  # noqa: E402
from bsvd_amd import evaluation  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-opt", required=True)
    ap.add_argument("--precision", default=None, choices=["fp32", "f16x3"])
    ap.add_argument("--force_yml", nargs="+", default=None, help="key:sub=value overrides")
    args = ap.parse_args()
    opt = yaml.safe_load(open(args.opt))
    for item in args.force_yml or []:
        keys, value = item.split("=")
        node = opt
        *path, last = keys.split(":")
        for k in path:
            node = node[k]
        node[last] = yaml.safe_load(value)
    opt["is_train"] = False
    seed = opt.get("manual_seed")
    if seed is not None:                       # basicsr.utils.set_random_seed
        random.seed(seed); np.random.seed(seed); torch.manual_seed(seed); torch.cuda.manual_seed_all(seed)
    net_opt = opt["network_g"]
    if net_opt["type"] not in bsvd_amd.ARCH_REGISTRY and net_opt["type"] + "_MI355X" in bsvd_amd.ARCH_REGISTRY:
        net_opt["type"] += "_MI355X"
    if args.precision:
        net_opt["precision"] = args.precision
    if net_opt.get("pretrain_ckpt") and not os.path.exists(net_opt["pretrain_ckpt"]):
        print("WARNING: %s not found -> random-init weights" % net_opt["pretrain_ckpt"])
        net_opt["pretrain_ckpt"] = None
    model = bsvd_amd.MODEL_REGISTRY.get(opt.get("model_type", "DenoisingModel"))(opt)
    metrics = (opt.get("val") or {}).get("metrics") or {}
    results = {}
    for key, dopt in sorted(opt["datasets"].items()):
        if dopt.get("type", "ValFolderDataset") != "ValFolderDataset":
            continue
        if not os.path.isdir(dopt["valsetdir"]):
            print("skip %s: %s does not exist" % (dopt.get("name", key), dopt["valsetdir"]))
            continue
        ds = evaluation.ValFolderDataset(dopt)
        per_folder, total = evaluation.evaluate(model, ds, metrics)
        results[dopt.get("name", key)] = {"folders": per_folder, "mean": total}
        print("%s: %s" % (dopt.get("name", key), json.dumps(total)))
    print(json.dumps(results, indent=1))


if __name__ == "__main__":
    main()
