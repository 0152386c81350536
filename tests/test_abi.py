"""CPU: the C-ABI shared library loads and exports every symbol include/tclight_hip.h declares."""
import ctypes
import os

from tc_light_amd.lib import LIB_PATH, parse_header


def test_exports_match_header():
    import __graft_entry__ as g
    if not os.path.exists(LIB_PATH):
        g.build()
    sig = parse_header()
    assert len(sig) >= 12
    dll = ctypes.CDLL(LIB_PATH)
    for name in sig:
        assert hasattr(dll, name), f"{name} declared in the header but not exported"


def test_pure_host_entry_points():
    from tc_light_amd.lib import lib
    L = lib()
    assert L.tcl_stage_workspace_bytes(16, 720, 960) > 2 * 16 * 3 * 720 * 960 * 4
    assert L.tcl_msssim_workspace_bytes(6, 176, 192) > 0


def test_product_never_imports_oracle():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for dp, _, fs in os.walk(os.path.join(root, "tc_light_amd")):
        for f in fs:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f
