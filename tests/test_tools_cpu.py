"""Development tooling that must not rot silently: the ablation branches of the flat k = 1 kernel live OUTSIDE the product source
(tools/exp_ablate.patch, applied to a scratch copy of csrc/grid.hip by tools/exp_ablate.sh) -- the patch has to keep applying."""
import shutil
import subprocess
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def test_ablation_patch_applies_to_the_product_kernel_source(tmp_path):
    src = ROOT / "pointcloudcomparator_amd" / "csrc" / "grid.hip"
    assert "PCC_ABLATE" not in src.read_text().replace("PCC_ABLATE bits", ""), "ablation code belongs in tools/exp_ablate.patch"
    work = tmp_path / "grid.hip"
    shutil.copy(src, work)
    p = subprocess.run(["patch", "-s", str(work), str(ROOT / "tools" / "exp_ablate.patch")], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout + p.stderr
    assert work.read_text().count("PCC_ABLATE") >= 8


def test_section_timing_patch_applies_to_the_knn_kernel_source(tmp_path):
    """tools/exp_knn_sections.patch (wave time per section of k_grid_knn_sel, a dev build) against csrc/knn.hip"""
    src = ROOT / "pointcloudcomparator_amd" / "csrc" / "knn.hip"
    assert "SECT(" not in src.read_text(), "the section timers belong in tools/exp_knn_sections.patch"
    work = tmp_path / "knn.hip"
    shutil.copy(src, work)
    p = subprocess.run(["patch", "-s", str(work), str(ROOT / "tools" / "exp_knn_sections.patch")], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout + p.stderr
    assert work.read_text().count("SECT(") >= 12
