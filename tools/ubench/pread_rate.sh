python3 - <<'PY'
import numpy as np
a = np.random.default_rng(1).integers(32, 120, size=1 << 30, dtype=np.uint8)
with open("/dev/shm/pr_test.bin", "wb") as f:
    for _ in range(3): f.write(a.tobytes())
PY
tools/ubench/pread_rate /dev/shm/pr_test.bin mmap
tools/ubench/pread_rate /dev/shm/pr_test.bin quick
tools/ubench/pread_rate /dev/shm/pr_test.bin
rm -f /dev/shm/pr_test.bin
