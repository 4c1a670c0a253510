#!/usr/bin/env python3
"""Subprocess driver of tests/test_plugin_boundary.py (TEST INFRASTRUCTURE; needs a fresh interpreter because the plug-in
registries are filled as import side effects and the reference's hard-coded ``.cuda()`` sites are patched process-wide).

  registry  <real|stub> <ref_first|engine_first>
        BasicSR's registry is made importable as ``basicsr.utils.registry`` -- ``real``: the reference's own
        BasicSR/basicsr/utils/registry.py loaded by path (container only); ``stub``: a strict stand-in with the same
        duplicate-name assertion -- and the reference plug-in (real: bsvd_arch.py, tsm_arch.py, denoising_model.py through
        the make_golden shim; stub: four dummy classes under the reference's names) is imported before / after
        ``bsvd_amd``.  Prints what each registry name resolves to before and after ``bsvd_amd.install``.
  drive_test
        container only: the reference's REAL ``DenoisingModel.test -> denoise_seq -> temp_denoise``
        (denoising_model.py:170-190, validation_seq_infer.py:10-100) drives (a) the reference ``BSVD`` and (b)
        ``bsvd_amd.BSVD`` with the CPU oracle executor patched in; prints the max-abs difference.
  drive_validation <tmpdir>
        container only: the reference's REAL ``DenoisingModel.validation`` (:192-367) with the reference ``BSVD`` against the
        engine's ``DenoisingModel.validation`` on a synthetic frame folder, both behind the calls
        ``basicsr.test_pipeline`` makes (BasicSR/basicsr/test.py:26-41); prints totals, log lines, CSVs.
"""
import contextlib
import json
import logging
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE, os.path.join(HERE, "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)
REF = os.environ.get("BSVD_REFERENCE", "/root/reference")

import numpy as np  # noqa: E402
import torch  # noqa: E402

NET = dict(chns=[16, 32, 64], mid_ch=16, shift_input=False, in_ch=4, out_ch=3, norm="none", act="relu6", interm_ch=16,
           blind=False, pretrain_ckpt=None)


def where(obj):
    return None if obj is None else obj.__module__.split(".")[0]


def stub_registry():
    """strict stand-in for basicsr.utils.registry: same surface, same duplicate-name assertion"""
    class Strict:
        def __init__(self, name):
            self._name, self._obj_map = name, {}

        def _do_register(self, name, obj):
            assert name not in self._obj_map, "An object named '%s' was already registered in '%s' registry!" % (name, self._name)
            self._obj_map[name] = obj

        def register(self, obj=None):
            if obj is None:
                def deco(o):
                    self._do_register(o.__name__, o)
                    return o
                return deco
            self._do_register(obj.__name__, obj)

        def get(self, name):
            if name not in self._obj_map:
                raise KeyError(name)
            return self._obj_map[name]

        def __contains__(self, name):
            return name in self._obj_map

        def keys(self):
            return self._obj_map.keys()

    for pkg in ("basicsr", "basicsr.utils"):
        m = types.ModuleType(pkg)
        m.__path__ = []
        sys.modules[pkg] = m
    r = types.ModuleType("basicsr.utils.registry")
    for k in ("DATASET", "ARCH", "MODEL", "LOSS", "METRIC"):
        setattr(r, k + "_REGISTRY", Strict(k.lower()))
    sys.modules["basicsr.utils.registry"] = r
    return r


def stub_reference_plugin(r):
    """what the reference's scans register (bsvd_arch.py:440, tsm_arch.py, denoising_model.py:15, video_dali_dataset.py:199)"""
    mod = types.ModuleType("Experimental_root")
    ns = {}
    for reg, names in ((r.ARCH_REGISTRY, ("BSVD", "TSN")), (r.MODEL_REGISTRY, ("DenoisingModel",)),
                       (r.DATASET_REGISTRY, ("ValFolderDataset",))):
        for n in names:
            cls = type(n, (), {"__module__": "Experimental_root"})
            reg.register()(cls)
            ns[n] = cls
    return mod


def real_reference_plugin():
    import make_golden as mg
    mg.import_reference()                 # basicsr.utils.registry (REAL file, by path) + bsvd_arch.py  -> BSVD
    mg.import_reference_tsn()             # tsm_arch.py                                                  -> TSN
    mg.import_reference_callers()         # denoising_model.py                                           -> DenoisingModel


def scenario_registry(kind, order):
    if kind == "stub":
        r = stub_registry()
        plug = lambda: stub_reference_plugin(r)              # noqa: E731
    else:
        import make_golden as mg
        for pkg in ("basicsr", "basicsr.utils"):
            m = types.ModuleType(pkg)
            m.__path__ = []
            sys.modules[pkg] = m
        mg._load("basicsr.utils.registry", os.path.join(REF, "BasicSR/basicsr/utils/registry.py"))
        load = mg._load                    # ONE registry module per process, as in a real run: the shim must not re-execute it
        mg._load = lambda name, path: sys.modules[name] if name == "basicsr.utils.registry" else load(name, path)
        plug = real_reference_plugin
    if order == "ref_first":
        plug()
        import bsvd_amd
    else:
        import bsvd_amd
        plug()
    from bsvd_amd import registry as R
    regs = sys.modules["basicsr.utils.registry"]
    assert R.HAVE_BASICSR and R.ARCH_REGISTRY is regs.ARCH_REGISTRY and R.MODEL_REGISTRY is regs.MODEL_REGISTRY \
        and R.DATASET_REGISTRY is regs.DATASET_REGISTRY

    def snapshot():
        out = {}
        for reg, names in ((regs.ARCH_REGISTRY, ("BSVD", "TSN")), (regs.MODEL_REGISTRY, ("DenoisingModel",)),
                           (regs.DATASET_REGISTRY, ("ValFolderDataset",))):
            for n in names:
                out[n] = where(reg._obj_map.get(n))
                out[n + "_MI355X"] = where(reg._obj_map.get(n + "_MI355X"))
        return out

    res = {"after_import": snapshot()}
    res["install_keep"] = bsvd_amd.install(replace=False)
    res["after_install_keep"] = snapshot()
    res["install_replace"] = bsvd_amd.install(replace=True)
    res["after_install_replace"] = snapshot()
    # the stock YAML's network_g / model_type now build the engine (basicsr/archs/__init__.py:19-25)
    net = bsvd_amd.build_network(dict(NET, type="BSVD"))
    res["built_arch"] = [type(net).__module__, type(net).__name__, isinstance(net, bsvd_amd.BSVD), net.shift_num]
    res["model_cls"] = regs.MODEL_REGISTRY.get("DenoisingModel") is bsvd_amd.DenoisingModel
    res["dataset_cls"] = regs.DATASET_REGISTRY.get("ValFolderDataset") is bsvd_amd.ValFolderDataset
    # a second install is idempotent, uninstall gives the reference its names back
    bsvd_amd.install(replace=True)
    bsvd_amd.uninstall()
    res["after_uninstall"] = snapshot()
    print("RESULT " + json.dumps(res))


# ------------------------------------------------------------------------------------------------------------------
def patch_engine_onto_oracle():
    """bsvd_amd's host logic on CPU: the executor the product hands its layers to is replaced by the oracle-backed one
    (tests/oracle_exec.py).  The product itself has no such path."""
    import bsvd_amd
    import bsvd_amd.arch as A
    import bsvd_amd.denoise as D
    from oracle_exec import OracleExecutor
    A._HipNet._device = lambda self: torch.device("cpu")
    A._HipNet._executor = lambda self, dev: OracleExecutor(dict(self._engine_state()))
    torch.cuda.device = lambda dev: contextlib.nullcontext()
    D.DenoisingModel._pick_device = staticmethod(lambda: torch.device("cpu"))
    return bsvd_amd


def reference_model(dm, net, opt):
    """the reference's DenoisingModel around ``net`` without BaseModel.__init__ (stubbed to ``object`` by the shim)"""
    m = object.__new__(dm.DenoisingModel)
    m.opt, m.net_g, m.device, m.is_train, m.center_frame_only = opt, net, torch.device("cpu"), False, False
    return m


def seeded_nets():
    import make_golden as mg
    ref = mg.import_reference()
    vsi, dm = mg.import_reference_callers()
    bsvd_amd = patch_engine_onto_oracle()
    rnet = ref.BSVD(**NET)
    st = mg.load_seeded(rnet, 21)
    enet = bsvd_amd.BSVD(**dict(NET, engine_mode="clip", precision="fp32"))
    enet.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in st.items()})
    return bsvd_amd, dm, rnet.eval(), enet.eval()


def scenario_drive_test():
    bsvd_amd, dm, rnet, enet = seeded_nets()
    from seeded import seeded_clip
    lq = torch.from_numpy(seeded_clip((1, 5, 3, 30, 50), 3, kind="sigma30"))[0]           # H, W not multiples of 4
    nm = torch.full((5, 1, 30, 50), 30.0 / 255.0)
    opt = {"val": {"temp_psz": -1}}
    outs = []
    for net in (rnet, enet):
        m = reference_model(dm, net, opt)
        m.feed_data({"lq": lq, "noise_map": nm})
        with torch.no_grad():
            m.test()                                  # REAL reference code: padding_input -> denoise_seq -> temp_denoise -> net
        outs.append(m.output)
    res = {"shape": list(outs[0].shape), "same_shape": outs[0].shape == outs[1].shape,
           "max_abs": float((outs[0] - outs[1]).abs().max()), "ref_range": [float(outs[0].min()), float(outs[0].max())],
           "engine_cls": type(enet).__module__, "ref_cls": type(rnet).__module__}
    print("RESULT " + json.dumps(res))


def make_folders(root):
    from PIL import Image
    rs = np.random.RandomState(7)
    for name, frames in (("bus", 4), ("car", 3)):
        d = os.path.join(root, "set", name)
        os.makedirs(d)
        base = rs.uniform(0, 255, (frames, 22, 34, 3))
        for i in range(frames):
            Image.fromarray(base[i].astype(np.uint8)).save(os.path.join(d, "%05d.png" % i))
    return os.path.join(root, "set")


def scenario_drive_validation(tmp):
    bsvd_amd, dm, rnet, enet = seeded_nets()
    from bsvd_amd import evaluation as E
    folders = make_folders(tmp)
    dopt = {"name": "synthetic_s30", "type": "ValFolderDataset", "valsetdir": folders, "num_validation_frames": 85,
            "valnoisestd": 30, "phase": "val"}
    opt = {"name": "run", "model_type": "DenoisingModel", "dist": False, "rank": 0, "is_train": False, "num_gpu": 1,
           "network_g": dict(NET, type="BSVD", engine_mode="clip", precision="fp32"),
           "path": {"visualization": os.path.join(tmp, "vis"), "log": tmp},
           "val": {"temp_psz": -1, "save_img": False,
                   "metrics": {"psnr": {"type": "calculate_psnr", "crop_border": 2},
                               "psnr_float": {"type": "calculate_psnr_float", "crop_border": 2}}}}
    bsvd_amd.install(replace=True)
    out = {}
    for who in ("reference", "engine"):
        # get_root_logger: a stream handler + a file handler (basicsr/utils/logger.py); the CSVs land next to handlers[1]
        log = logging.getLogger("basicsr")
        for h in list(log.handlers):
            log.removeHandler(h)
        log.setLevel(logging.INFO)
        log.propagate = False
        lines = []

        class Grab(logging.Handler):
            def emit(self, record):
                lines.append(record.getMessage())

        log.addHandler(Grab())
        logfile = os.path.join(tmp, "test_%s.log" % who)
        log.addHandler(logging.FileHandler(logfile))
        torch.manual_seed(10)                          # the AWGN comes from the global RNG (video_dali_dataset.py:229)
        # ---- the calls of basicsr.test_pipeline (BasicSR/basicsr/test.py:26-41) ----
        test_set = bsvd_amd.build_dataset(dopt)                                              # build_dataset
        test_set.device = torch.device("cpu")
        if who == "reference":
            # the reference squeezes val_data['lq'/'gt'] IN PLACE after feed_data (denoising_model.py:252-253).  On the GPU
            # ``gt.to('cuda')`` is a copy (the dataset keeps gt on the host, video_dali_dataset.py:224), so self.gt stays
            # 5-D; with everything on the CPU ``.to`` would alias it.  Hand out a fresh gt per access to keep that copy.
            class HostGt(dict):
                def __getitem__(self, k):
                    v = dict.__getitem__(self, k)
                    return v.clone() if k == "gt" else v

            get = type(test_set).__getitem__
            test_set = type("RefViewDataset", (type(test_set),), {"__getitem__": lambda self, i: HostGt(get(self, i))})(dopt)
            test_set.device = torch.device("cpu")
        loader = torch.utils.data.DataLoader(test_set, batch_size=1, shuffle=False, num_workers=0)   # build_dataloader ('val')
        if who == "engine":
            model = bsvd_amd.build_model(opt)                                                # build_model
            model.net_g.load_state_dict(enet.state_dict())
        else:
            dm.get_root_logger = lambda *a, **k: log
            dm.tensor2img = lambda ts: E.tensor2img(ts[0])
            dm.calculate_metric = lambda data, mo: E.METRICS[mo["type"]](**data, **{k: v for k, v in mo.items() if k != "type"})
            dm.imwrite = E.imwrite
            # build_model constructs the network between the seeding and the first noise draw; the constructor consumes the
            # global RNG (Kaiming init, bsvd_arch.py:476-483) -- identically in both classes (golden g11)
            type(rnet)(**NET)
            model = reference_model(dm, rnet, opt)
        total = model.validation(loader, current_iter=opt["name"], tb_logger=None, save_img=opt["val"]["save_img"])
        csvs = {}
        for f in sorted(os.listdir(tmp)):
            if f.startswith("test_%s" % who) and f.endswith(".csv"):
                csvs[f.replace("test_%s" % who, "")] = open(os.path.join(tmp, f)).read()
        out[who] = {"total": total, "log": [l for l in lines if l.startswith("Validation")], "csv": csvs,
                    "model": type(model).__module__, "net": type(model.net_g).__module__}
    print("RESULT " + json.dumps(out))


if __name__ == "__main__":
    if sys.argv[1] == "registry":
        scenario_registry(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "drive_test":
        scenario_drive_test()
    elif sys.argv[1] == "drive_validation":
        scenario_drive_validation(sys.argv[2])
    else:
        raise SystemExit("unknown scenario")
