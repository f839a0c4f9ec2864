"""No-GPU checks of the drop-in boundary: the C-ABI library loads, exports every symbol include/nuslam_hip.h
declares, its pure host helpers agree with the oracle, and it refuses to run without a device (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import _oracle as O
import nuslam_hip as nh

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "nuslam_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nuslam_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = C.CDLL(nh.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), "libnuslam_hip.so does not export " + n
    assert sorted(s[0] for s in nh.SYMBOLS) == names, "python binding table and header disagree"
    assert nh.lib().nuslam_abi_version() == 3


def test_host_helpers_match_oracle():
    rng = np.random.default_rng(3)
    for _ in range(50):
        x, y = rng.normal(size=2)
        assert np.array_equal(nh.cartesian2polar(x, y), O.cartesian2polar(x, y))
    s = rng.normal(size=3 + 2 * 6)
    for j in range(1, 7):
        assert np.array_equal(nh.measurement(s, j), O.measurement(s, j))
        assert np.array_equal(nh.jacobian(s, j), O.jacobian(s, j))
    with pytest.raises(nh.NuslamError) as ei:
        nh.measurement(s, 7)                      # landmark index out of bounds
    assert ei.value.code == nh.E_BOUNDS


def test_no_cpu_fallback():
    if nh.device_count() > 0:
        pytest.skip("a GPU is visible: the refusal path is for device-less hosts")
    with pytest.raises(nh.NuslamError) as ei:
        nh.EKF(np.zeros(3), np.zeros(4), np.eye(3), np.eye(2))
    assert ei.value.code == nh.E_NODEV
    with pytest.raises(nh.NuslamError):
        nh.Batch(2, 3, np.eye(3), np.eye(2))


def test_error_strings():
    L = nh.lib()
    for code in range(9):
        assert L.nuslam_strerror(code)
    assert b"logic_error" in L.nuslam_strerror(nh.E_BOUNDS)
    assert b"runtime_error" in L.nuslam_strerror(nh.E_SINGULAR)


def test_build_info_names_the_kernel_sources():
    info = nh.build_info()
    assert re.fullmatch(r"csrc=[0-9a-f]{16}", info), info


def test_rccl_reduction_refuses_without_a_device():
    """The batch reduction binds RCCL at run time; on a host without a GPU it must fail loudly, not pretend."""
    if nh.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(nh.NuslamError) as ei:
        nh.Comm(bytes(nh.COMM_ID_BYTES), 1, 0, 0)
    assert ei.value.code in (nh.E_NODEV, nh.E_COMM)


def test_generated_strip_blocks_are_in_step_with_their_generator(tmp_path):
    """csrc/ekf_strips_blocks.inc (the inline-asm coefficient blocks of k_tick_strips_lane) is what tools/gen_strips_asm.py writes."""
    import importlib.util, os, shutil
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    inc = os.path.join(root, "shermbot-navigation_amd", "csrc", "ekf_strips_blocks.inc")
    committed = open(inc).read()
    # run the generator on a copy of the tree layout it expects
    work = tmp_path / "w"
    (work / "tools").mkdir(parents=True)
    (work / "shermbot-navigation_amd" / "csrc").mkdir(parents=True)
    shutil.copy(os.path.join(root, "tools", "gen_strips_asm.py"), work / "tools" / "gen_strips_asm.py")
    spec = importlib.util.spec_from_file_location("gen_strips_asm_copy", str(work / "tools" / "gen_strips_asm.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert open(work / "shermbot-navigation_amd" / "csrc" / "ekf_strips_blocks.inc").read() == committed
