"""In-tree build of libloupiote_hip.so (hipcc --offload-arch=gfx950; cross-compiles without a GPU)."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))


def build(force=False):
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j8", "-s"]
    if force:
        cmd.append("-B")
    subprocess.check_call(cmd)
    so = os.path.join(_HERE, "libloupiote_hip.so")
    if not os.path.exists(so):
        raise RuntimeError("build did not produce " + so)
    return so
