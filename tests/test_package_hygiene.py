"""Package hygiene the CPU suite can check without a GPU: every global name the package's functions read is bound (the GPU-only
paths of the engine never run here: a missing import there would first show on the GPU box), and no module of the product
package is a monolith again."""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_global_name_is_bound():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "check_names.py")], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout


def test_the_checker_has_teeth(tmp_path):
    src = open(os.path.join(ROOT, "lanpaint_amd", "capture.py")).read().replace("from .masks import _compact_mask\n", "")
    f = tmp_path / "broken.py"
    f.write_text(src)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "check_names.py"), str(f)], capture_output=True, text=True)
    assert p.returncode == 1 and "_compact_mask" in p.stdout


def test_no_module_of_the_product_package_exceeds_900_lines():
    """lanpaint.py was 2 029 lines with seven launch modes in one class.  engine.py (one sigma call as
    launches), capture.py (the capture / replay state machine), loops.py (loops off the fast path), masks.py, buffers.py."""
    sizes = {os.path.basename(f): sum(1 for _ in open(f)) for f in glob.glob(os.path.join(ROOT, "lanpaint_amd", "*.py"))}
    assert max(sizes.values()) <= 900, sorted(sizes.items(), key=lambda kv: -kv[1])[:3]
    assert {"engine.py", "capture.py", "loops.py", "masks.py", "buffers.py", "lanpaint.py"} <= set(sizes)
    assert sum(1 for _ in open(os.path.join(ROOT, "bench.py"))) <= 350


def test_engine_class_is_assembled_from_its_parts():
    from lanpaint_amd import LanPaint
    from lanpaint_amd.capture import GraphReplay
    from lanpaint_amd.engine import EngineCore
    from lanpaint_amd.loops import ThinkLoops
    assert LanPaint.__mro__[1:4] == (GraphReplay, ThinkLoops, EngineCore)
    assert LanPaint._OWN_METHODS == (LanPaint.langevin_dynamics, LanPaint.score_model, LanPaint.prepare_step_size)

    class Sub(LanPaint):
        def score_model(self, *a, **k):
            return super().score_model(*a, **k)
    probe = object.__new__(Sub)
    assert probe._overridden("score_model") and not probe._overridden("langevin_dynamics")
    assert not object.__new__(LanPaint)._overridden("score_model")
