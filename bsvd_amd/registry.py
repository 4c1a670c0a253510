"""Plug-in registries: the drop-in boundary of the reference is BasicSR's ARCH_REGISTRY / MODEL_REGISTRY
(/root/reference/BasicSR/basicsr/utils/registry.py:4-82, used at
/root/reference/Experimental_root/archs/bsvd_arch.py:440 and BasicSR/basicsr/archs/__init__.py:19-25).

If BasicSR is importable its registries are used, so ``network_g: {type: BSVD}`` in a stock YAML resolves to
this engine; otherwise a minimal compatible registry is provided so the same code runs stand-alone.
"""


class Registry:
    """name -> class map with decorator registration (same surface as BasicSR's Registry)."""

    def __init__(self, name):
        self._name = name
        self._obj_map = {}

    def register(self, obj=None, name=None, replace=False):
        def do(o):
            key = name or o.__name__
            if key in self._obj_map and not replace:
                raise AssertionError("An object named '%s' was already registered in '%s' registry!" % (key, self._name))
            self._obj_map[key] = o
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
    from basicsr.utils.registry import ARCH_REGISTRY, MODEL_REGISTRY  # type: ignore
    HAVE_BASICSR = True
except Exception:  # noqa: BLE001  (basicsr needs cv2/torchvision/version.py; any failure -> stand-alone)
    ARCH_REGISTRY = Registry("arch")
    MODEL_REGISTRY = Registry("model")
    HAVE_BASICSR = False


def register_arch(cls):
    """Registers under cls.__name__ unless that name is already taken (e.g. by the reference's own BSVD,
    whose registry asserts on duplicates, registry.py:38-41); then ``<name>_MI355X`` is used."""
    name = cls.__name__
    if name in ARCH_REGISTRY:
        name = name + "_MI355X"
    if name not in ARCH_REGISTRY:
        if HAVE_BASICSR:
            ARCH_REGISTRY._do_register(name, cls)
        else:
            ARCH_REGISTRY.register(cls, name=name)
    return cls


def build_network(opt):
    """basicsr.archs.build_network: pops ``type`` and instantiates the registered class with the rest."""
    opt = dict(opt)
    net_type = opt.pop("type")
    return ARCH_REGISTRY.get(net_type)(**opt)
