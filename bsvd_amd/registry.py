"""Plug-in registries: the drop-in boundary of the reference is BasicSR's ARCH_REGISTRY / MODEL_REGISTRY / DATASET_REGISTRY
(/root/reference/BasicSR/basicsr/utils/registry.py:4-82; filled as import side effects by
Experimental_root/archs/__init__.py:5-9 -> bsvd_arch.py:440, tsm_arch.py; models/__init__.py:5-9 -> denoising_model.py:15;
data/__init__.py -> video_dali_dataset.py:199; consumed by basicsr/archs/__init__.py:19-25, basicsr/models/__init__.py:19-30,
basicsr/data/__init__.py:25).

BasicSR's registry ASSERTS on a duplicate name (registry.py:38-41) and the reference's scans register ``BSVD``, ``TSN``,
``DenoisingModel`` and ``ValFolderDataset``.  So, when the real registries are importable, importing ``bsvd_amd`` never
touches those four names -- in either import order nothing can collide: the engine's classes are registered under
``<name>_MI355X`` only.  ``bsvd_amd.install()`` is the explicit step that puts the engine under the STOCK names:

    import bsvd_amd
    bsvd_amd.install(replace=True)     # stock YAMLs (type: BSVD, model_type: DenoisingModel, ...) now build the engine

``install`` first imports the reference's plug-in packages if they are importable (so that their registrations have
happened and cannot assert later), then fills free stock names and -- with ``replace=True`` -- swaps the entries the
reference holds; ``uninstall()`` restores them.  Without BasicSR a compatible stand-alone registry is provided and the
engine owns the stock names from the start (nobody else can claim them).
"""
import importlib

SUFFIX = "_MI355X"


class Registry:
    """name -> class map with decorator registration (same surface as BasicSR's Registry)."""

    def __init__(self, name):
        self._name = name
        self._obj_map = {}

    def _do_register(self, name, obj):
        if name in self._obj_map:
            raise AssertionError("An object named '%s' was already registered in '%s' registry!" % (name, self._name))
        self._obj_map[name] = obj

    def register(self, obj=None, name=None):
        def do(o):
            self._do_register(name or o.__name__, o)
            return o

        return do if obj is None else do(obj)

    def get(self, name):
        if name not in self._obj_map:
            raise KeyError("No object named '%s' found in '%s' registry!" % (name, self._name))
        return self._obj_map[name]

    def __contains__(self, name):
        return name in self._obj_map

    def __iter__(self):
        return iter(self._obj_map.items())

    def keys(self):
        return self._obj_map.keys()


try:  # real BasicSR present -> plug into it
    from basicsr.utils.registry import ARCH_REGISTRY, DATASET_REGISTRY, MODEL_REGISTRY  # type: ignore
    HAVE_BASICSR = True
except Exception:  # noqa: BLE001  (basicsr needs cv2/torchvision/version.py; any failure -> stand-alone)
    ARCH_REGISTRY = Registry("arch")
    MODEL_REGISTRY = Registry("model")
    DATASET_REGISTRY = Registry("dataset")
    HAVE_BASICSR = False

_REGS = {"arch": ARCH_REGISTRY, "model": MODEL_REGISTRY, "dataset": DATASET_REGISTRY}
_ENGINE = {"arch": {}, "model": {}, "dataset": {}}       # kind -> {stock name: engine class}
_REPLACED = {}                                            # (kind, name) -> the class install(replace=True) displaced
REFERENCE_PLUGINS = ("Experimental_root.archs", "Experimental_root.models", "Experimental_root.data")


def _register(kind, cls):
    reg, name = _REGS[kind], cls.__name__
    _ENGINE[kind][name] = cls
    reg._obj_map[name + SUFFIX] = cls                     # always there, whatever the import order
    if not HAVE_BASICSR:
        reg._obj_map[name] = cls                          # stand-alone registry: the stock names are ours
    return cls


def register_arch(cls):
    """``<name>_MI355X`` in ARCH_REGISTRY (+ the stock name in the stand-alone registry; see ``install``)."""
    return _register("arch", cls)


def register_model(cls):
    return _register("model", cls)


def register_dataset(cls):
    return _register("dataset", cls)


def install(replace=False, import_reference=True):
    """Puts the engine's classes under the reference's stock names (``BSVD``, ``TSN``, ``DenoisingModel``,
    ``ValFolderDataset``).  Free names are always taken; names the reference plug-in holds are swapped only with
    ``replace=True`` (the registry entry is exchanged, nothing asserts).  ``import_reference``: import the reference's
    plug-in packages first when they are importable, so their import-time registrations cannot collide afterwards
    (import-order caveat: a reference plug-in imported AFTER ``install()`` took a free stock name trips BasicSR's duplicate
    assertion at its own ``register()`` -- call ``uninstall()`` first, or let ``install`` do the import, which is the default)
    (``Experimental_root.data`` needs NVIDIA DALI and normally fails on ROCm -- then ``ValFolderDataset`` is simply free).
    Returns {kind: {name: 'engine' | 'reference'}}: who answers to each stock name now."""
    if import_reference and HAVE_BASICSR:
        for mod in REFERENCE_PLUGINS:
            try:
                importlib.import_module(mod)
            except Exception:  # noqa: BLE001  (absent, or a CUDA-only dependency such as nvidia.dali)
                pass
    report = {}
    for kind, reg in _REGS.items():
        report[kind] = {}
        for name, cls in _ENGINE[kind].items():
            cur = reg._obj_map.get(name)
            if cur is None or cur is cls:
                reg._obj_map[name] = cls
            elif replace:
                _REPLACED.setdefault((kind, name), cur)
                reg._obj_map[name] = cls
            report[kind][name] = "engine" if reg._obj_map[name] is cls else "reference"
    return report


def uninstall():
    """Undoes ``install``: displaced reference classes return to their names; names the engine had taken because they
    were free are released again (plugged into BasicSR only -- the stand-alone registry keeps them)."""
    for kind, reg in _REGS.items():
        for name, cls in _ENGINE[kind].items():
            if (kind, name) in _REPLACED:
                reg._obj_map[name] = _REPLACED.pop((kind, name))
            elif HAVE_BASICSR and reg._obj_map.get(name) is cls:
                del reg._obj_map[name]


def engine_name(kind, name):
    """The registry key under which the ENGINE's class for stock ``name`` can be built right now."""
    reg, cls = _REGS[kind], _ENGINE[kind].get(name)
    if cls is not None and reg._obj_map.get(name) is not cls and name + SUFFIX in reg:
        return name + SUFFIX
    return name


def _lookup(kind, name):
    """The class ``bsvd_amd.build_*`` instantiates for ``name``: whatever the registry holds under it -- and, when BasicSR is
    importable, ``install()`` has not run and the stock name is therefore still FREE (the engine only registers
    ``<name>_MI355X`` so as not to squat on the reference's names), the engine's own class for that stock name.  The same
    ``bsvd_amd.build_network({'type': 'BSVD'})`` thus works stand-alone and plugged in; a name the REFERENCE holds is never
    shadowed here (that takes ``install(replace=True)``)."""
    reg = _REGS[kind]
    if name not in reg._obj_map and name in _ENGINE[kind]:
        return _ENGINE[kind][name]
    return reg.get(name)


def build_network(opt):
    """basicsr.archs.build_network (basicsr/archs/__init__.py:19-25): pops ``type`` and instantiates the registered class
    with the rest."""
    opt = dict(opt)
    net_type = opt.pop("type")
    return _lookup("arch", net_type)(**opt)


def build_dataset(dataset_opt):
    """basicsr.data.build_dataset (basicsr/data/__init__.py:25)."""
    return _lookup("dataset", dataset_opt["type"])(dict(dataset_opt))


def build_model(opt):
    """basicsr.models.build_model (basicsr/models/__init__.py:19-30)."""
    return _lookup("model", opt["model_type"])(dict(opt))
