"""The experiments flavour of the library (libsdfhip_lab.so, include/sdfhip_experimental.h) beside the product in one process.

    import sdfbox_amd.lab
    sbx = sdfbox_amd.lab.load()        # this package again, bound to libsdfhip_lab.so: sbx.Scene, sbx.TUNE_ONE_KERNEL, ...

Used by the tests (the A/B kernel forms, the superseded gather formats and the test hooks stay regression-tested against the
oracle) and by the A/B scripts under scripts/.  A host application never needs it."""
import importlib.util
import os
import sys

_NAME = "sdfbox_amd_lab"


def load():
    if _NAME in sys.modules:
        return sys.modules[_NAME]
    here = os.path.dirname(os.path.abspath(__file__))
    lib = os.path.join(here, "libsdfhip_lab.so")
    if not os.path.exists(lib):
        raise ImportError(f"{lib} is missing: `make -C {os.path.join(here, 'csrc')}` builds both flavours")
    saved = os.environ.get("SDFHIP_LIB")
    os.environ["SDFHIP_LIB"] = lib
    try:
        spec = importlib.util.spec_from_file_location(_NAME, os.path.join(here, "__init__.py"), submodule_search_locations=[here])
        mod = importlib.util.module_from_spec(spec)
        sys.modules[_NAME] = mod
        try:
            spec.loader.exec_module(mod)
        except BaseException:
            # a failed import leaves nothing behind: the package AND the submodules it had imported so far (sdfbox_amd_lab._lib ...),
            # or a second load() would meet a half-initialised package
            for name in [k for k in sys.modules if k == _NAME or k.startswith(_NAME + ".")]:
                del sys.modules[name]
            raise
    finally:
        if saved is None:
            del os.environ["SDFHIP_LIB"]
        else:
            os.environ["SDFHIP_LIB"] = saved
    assert mod._lib.EXPERIMENTS
    return mod
