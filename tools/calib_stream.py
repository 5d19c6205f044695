"""GPU box, under rocprofv3 --pmc: streams 2^28 doubles (2 GiB each way, far beyond the Infinity Cache) three times
through k_calibration_stream, the library's 8-byte-per-lane access width (see outfit_hip_calibration_stream)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from outfit_amd import _lib  # noqa: E402

N = 1 << 28
rc = _lib.lib().outfit_hip_calibration_stream(N, 3)
assert rc == 0, _lib.last_error()
print(f"calibration: {N} doubles x 3 launches, {N * 8} bytes read and {N * 8} bytes written per launch")
