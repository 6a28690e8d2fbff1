"""csrc/libm_f32.hpp -- glibc 2.35's sinf / cosf / atan2f restated for the device -- against the host's libm, bit for bit
(tests/cpp/test_libm.cpp): the oracle's pcl::eigen33 calls the host libm as PCL does (std::atan2 / cos / sin on floats,
reference src/segmentation.cpp:232-241); the device must take the same roots, or normals and curvatures differ in their last
bits (2.3 % of all points did until round 5).  No GPU."""
import subprocess
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def test_restated_sinf_cosf_atan2f_carry_the_host_libms_bits():
    subprocess.check_call(["make", "build/test_libm"], cwd=ROOT)
    r = subprocess.run([str(ROOT / "build" / "test_libm")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "libm ok" in r.stdout, r.stdout[-2000:]
    assert " 0 mismatches" in r.stdout
