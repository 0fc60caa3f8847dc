"""Hash of the implicit-GEMM kernel sources (shared by tools/pmc_bench_traffic.py, which stamps the PMC summary with it, and
bench.py, which drops `roofline.traffic` when the stamp does not match the sources it runs)."""
import hashlib
from pathlib import Path


def gemm_source_hash() -> str:
    """Hash of the implicit-GEMM kernel sources these counters describe: bench.py drops `roofline.traffic` when it differs
    from the sources it runs (a committed PMC summary must not outlive the kernels it was measured on)."""
    src = Path(__file__).resolve().parents[1] / "geo-deep-learning_amd" / "csrc"
    h = hashlib.sha256()
    for f in sorted(list(src.glob("conv_gemm*")) + list(src.glob("conv3x3_*.hip"))):
        h.update(f.name.encode())
        h.update(f.read_bytes())
    return h.hexdigest()[:16]
