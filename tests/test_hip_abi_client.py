"""The C ABI without PyTorch: tests/abi_client/abi_client.cpp links libgkg_hip.so through include/gkg_hip.h only, takes its device
memory from the HIP runtime and checks gkg_knn_fwd / gkg_mr_fwd / gkg_mr_bwd (the drop-in calls for torch_edge.py:164-176 and
torch_vertex.py:49-54) bit for bit against the C oracle on five cases.  The program is built by __graft_entry__.build()."""
import os
import subprocess

import pytest

EXE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "abi_client", "abi_client")


def _exe():
    if not os.path.exists(EXE):
        import __graft_entry__ as g
        g.build()                        # (library, oracle, then the client: the link needs both shared objects)
    return EXE


def test_client_links_against_the_library_and_the_header_only():
    """CPU: the program loads (both shared objects resolve through its rpath) and says that there is no device."""
    p = subprocess.run([_exe()], capture_output=True, text=True, timeout=120)
    import torch
    if not torch.cuda.is_available():
        assert p.returncode == 3 and "no HIP device" in p.stderr, (p.returncode, p.stderr[-500:])
    out = subprocess.run(["ldd", _exe()], capture_output=True, text=True).stdout
    assert "libgkg_hip.so" in out and "not found" not in out, out
    assert "libtorch" not in out and "libc10" not in out, out                      # no PyTorch anywhere in the client


@pytest.mark.gpu
def test_client_is_bit_exact_against_the_oracle():
    p = subprocess.run([_exe()], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-2000:])
    assert "all 5 cases bit-exact against the oracle" in p.stdout, p.stdout[-2000:]
