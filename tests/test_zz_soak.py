"""Stress, not parity: tools/soak.py in small, one test per calling mode.  This file sorts LAST on purpose -- under `pytest -x`
every deterministic parity test has run before a soak can fail (in round 2 a soak in the middle of the collection order
cost the driver's record 820 tests).  Each mode sends the same batches through the asynchronous forms of one handle, round
after round on several HIP streams, and compares every result with the first blocking one of that batch."""
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
MODES = {  # mode -> result lines expected
    "gpt2": 1,     # fused RegexSplit + BPETokenizer: device two-half calls on three streams, pinned-host calls on four
    "llama3": 1,
    "bert": 1,     # the fused BERT chain (WordPiece)
    "detok": 1,    # the fused detokenizer
    "small": 2,    # encode_small_kernel (one launch), two tokenizers
    "wire": 1,     # the encode straight into an exchange wire (compact_kernel<WireSink>)
    "ops": 1,      # blocking ops taking turns on the pooled workspaces
}


@pytest.mark.gpu
@pytest.mark.parametrize("mode", list(MODES))
def test_soak(hip_lib, mode):
    p = subprocess.run([sys.executable, str(ROOT / "tools" / "soak.py"), "120", mode], capture_output=True, text=True, timeout=600,
                       cwd=str(ROOT))
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if " rounds " in ln and "MISMATCH" not in ln]
    assert len(lines) == MODES[mode] and all(ln.endswith("bad 0") for ln in lines), p.stdout[-2000:]
