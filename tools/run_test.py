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
    ap.add_argument("--csv_dir", default=None, help="write one <dataset>_<folder>.csv of per-frame metrics per folder, columns "
                                                    "<folder>_<metric index> like the reference (denoising_model.py:335-345)")
    ap.add_argument("--save_img", default=None, metavar="DIR", help="write the denoised frames as "
                                                                    "DIR/<dataset>/<folder>/<idx:08d>_<name>.png (val.save_img)")
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
    bsvd_amd.install(replace=True)             # the stock names of the YAML (BSVD / TSN / DenoisingModel) -> the engine
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
        dname = dopt.get("name", key)
        per_frame = {}
        per_folder, total = evaluation.evaluate(model, ds, metrics, per_frame=per_frame, run_name=opt.get("name", "bsvd"),
                                                save_img_dir=os.path.join(args.save_img, dname) if args.save_img else None)
        results[dname] = {"folders": per_folder, "mean": total}
        # the reference's log line (denoising_model.py:353-359)
        log = "Validation %s\n" % dname
        for mi, (metric, value) in enumerate(total.items()):
            log += "\t # %s: %.4f" % (metric, value) + "".join("\t # %s: %.4f" % (f, v[metric]) for f, v in per_folder.items()) + "\n"
        print(log, end="")
        if args.csv_dir:
            os.makedirs(args.csv_dir, exist_ok=True)
            for folder, acc in per_frame.items():
                with open(os.path.join(args.csv_dir, "%s_%s.csv" % (dname, folder)), "w") as fh:
                    cols = list(acc)
                    fh.write("," + ",".join("%s_%d" % (folder, i) for i in range(len(cols))) + "\n")
                    for r in range(len(acc[cols[0]])):
                        fh.write("%d," % r + ",".join(repr(float(acc[c][r])) for c in cols) + "\n")
    print(json.dumps(results, indent=1))


if __name__ == "__main__":
    main()
