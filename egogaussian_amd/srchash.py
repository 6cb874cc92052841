"""Prints the hash of the library's sources (what `make` embeds into libegs_raster.so and lib.kernel_source_hash() recomputes)."""
import glob
import hashlib
import os

_HERE = os.path.dirname(os.path.abspath(__file__))


def source_hash():
    """sha256 (first 16 hex digits) over csrc/*.hip, csrc/*.h, csrc/Makefile in name order, then include/egs_raster.h."""
    csrc = os.path.join(_HERE, "csrc")
    files = sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h")) + [os.path.join(csrc, "Makefile")])
    files.append(os.path.join(os.path.dirname(_HERE), "include", "egs_raster.h"))
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(source_hash())
