"""CPU: the C-ABI library loads and exports every symbol include/mmd_amd.h declares (no compute without a GPU)."""
import os
import re

from mmd_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols(header="mmd_amd.h"):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mmd_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/mmd_amd.h but not exported"
    assert sorted(_lib.EXPORTED_SYMBOLS) == declared, "ctypes signature table and header disagree"
    assert lib.mmd_abi_version() == _lib.ABI_VERSION
    debug = _declared_symbols("mmd_amd_debug.h")
    for name in debug:
        assert hasattr(lib, name), f"{name} declared in include/mmd_amd_debug.h but not exported"
    assert sorted(_lib.DEBUG_SYMBOLS) == debug
    assert not set(debug) & set(declared), "measurement hooks must stay out of the product header"


def test_unet_spec_matches_library():
    from mmd_amd.unet_spec import unet_param_spec
    import numpy as np
    lib = _lib.load()
    spec = unet_param_spec(4, 32, (1, 2, 4))
    assert lib.mmd_unet_num_tensors(32, 3) == len(spec) == 148
    # every doubling ladder the reference's constructor admits at these widths, UNET_DIM_MULTS[1] = (1, 2, 4, 8) among them
    for uid in (8, 16, 32, 64):
        for dm in ((1,), (1, 2), (1, 2, 4), (1, 2, 4, 8)):
            spec = unet_param_spec(4, uid, dm)
            assert lib.mmd_unet_num_tensors(uid, len(dm)) == len(spec)
            for i, shape in enumerate(spec.values()):
                assert lib.mmd_unet_tensor_numel(uid, len(dm), i) == int(np.prod(shape)), (uid, dm, i)
    for uid, nl in ((12, 3), (128, 3), (32, 5), (32, 0)):
        assert lib.mmd_unet_num_tensors(uid, nl) == -1 and b"unsupported" in lib.mmd_last_error()
