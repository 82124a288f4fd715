"""BUILD-CONTAINER tests (skipped wherever /root/reference does not exist, e.g. on the GPU box): what the repo says about the reference's own
code is re-derived from it.
  * the committed tests/golden/ref_host_*.npz ARE what scripts/make_ref_fixtures.py produces from /root/reference today (regenerated into a
    temporary directory and compared entry by entry);
  * scripts/check_oracle_vs_reference_text.py: the reference's integrator_euler.py, imported unchanged over a stand-in of the `wp` builtins
    and stepped serially, agrees with oracle/ref_torch.py on the golden rollouts (a stand-in pins nothing; it checks the transcription)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("PPR_REFERENCE", "/root/reference")
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "diffphys")), reason="needs the reference checkout (build container only)")


def test_committed_reference_fixtures_are_reproducible(tmp_path):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "make_ref_fixtures.py"), "--out", str(tmp_path)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    for name in ("ref_host_reduce_loss.npz", "ref_host_small.npz", "ref_host_mocap.npz", "ref_host_timemlp.npz"):
        with np.load(os.path.join(ROOT, "tests", "golden", name)) as a, np.load(os.path.join(str(tmp_path), name)) as b:
            assert sorted(a.files) == sorted(b.files), name
            for k in a.files:
                x, y = a[k], b[k]
                if x.dtype.kind in "fc":
                    assert x.shape == y.shape and np.array_equal(x, y, equal_nan=True), (name, k)
                else:
                    assert np.array_equal(x, y), (name, k)


def test_oracle_follows_the_reference_text():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "check_oracle_vs_reference_text.py")], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "the restatement follows the reference's text" in out.stdout
    assert out.stdout.count("contacts active") == 4   # Laikago, human, quad, the toy robot with a FIXED joint: ground contacts loaded in each


def test_phys_model_forward_fixture_is_reproducible(tmp_path):
    """tests/golden/ref_text_phys_model_forward.npz IS what the reference's phys_model.forward / backward text gives over the stand-ins today"""
    # (4 threads: with the default -- one per core for torch AND for the C oracle's OpenMP -- the two pools fight and the run takes 2-10 x longer)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "check_phys_model_vs_reference_text.py"), "--out", str(tmp_path / "f.npz")],
                         capture_output=True, text=True, timeout=900, env=dict(os.environ, OMP_NUM_THREADS="4"))
    assert out.returncode == 0, out.stderr[-2000:]
    with np.load(os.path.join(ROOT, "tests", "golden", "ref_text_phys_model_forward.npz")) as a, np.load(str(tmp_path / "f.npz")) as b:
        assert sorted(a.files) == sorted(b.files)
        for k in a.files:
            if a[k].dtype.kind == "f":
                assert np.allclose(a[k], b[k], rtol=1e-9, atol=1e-15), k   # (float64 oracle on a multi-threaded torch: sums may reorder)
            else:
                assert np.array_equal(a[k], b[k]), k


def test_import_urdf_follows_the_reference_text():
    """row f1: the reference's parse_urdf, imported unchanged over an urdfpy adapter on this package's XML reader and this package's
    ModelBuilder, fills the builder exactly as diffphys_amd.import_urdf does -- every URDF the reference ships, floating / fixed base,
    URDF inertials / density (scripts/check_import_urdf_vs_reference_text.py)"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "check_import_urdf_vs_reference_text.py")], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "SAME builders" in out.stdout and out.stdout.count(".urdf") >= 13
