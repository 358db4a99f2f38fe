"""Hash of the consensus kernel's sources: stamps measurements that belong to one build (profiles/traffic.json)."""
import hashlib
import os

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def _files():
    d = os.path.join(_ROOT, "racon_amd", "csrc")
    # the consensus kernel's sources (the window-construction and pair-alignment kernels live in the same library but do not
    # touch the traffic of poa_window_kernel2)
    return sorted(f for f in os.listdir(d) if (f.startswith("poa_") and f.endswith((".hpp", ".inc"))) or f in ("engine.hip", "engine_deep.hip", "engine_small.hip", "Makefile"))


def kernel_source_hash() -> str:
    h = hashlib.sha256()
    for f in _files():
        with open(os.path.join(_ROOT, "racon_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(kernel_source_hash())
