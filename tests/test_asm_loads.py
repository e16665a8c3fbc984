"""mcmc_chain_tab's rejected-step loop issues its LDS loads from inline asm and waits for them by hand: the
compiler's scoreboard does not see them, so the ASSEMBLY is checked -- no instruction may touch a row register between its load
and the wait that covers it, in any instantiation.  Needs hipcc only (cross-compiles without a GPU)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_hand_issued_lds_loads_are_waited_for_before_use():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "check_asm_loads.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "violations: 0" in r.stdout
