"""The C ABI used from plain C (-m gpu): tests/native/capi_smoke.c is compiled as strict C99 against include/kidmp.h,
linked with libkidmp.so, and must leave exactly the state the Python wrapper gets from the same inputs."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_program_equals_python_wrapper(tmp_path, gpu_warm):
    exe = str(tmp_path / "capi_smoke")
    lib = os.path.join(ROOT, "kid_amd")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "tests", "native", "capi_smoke.c"), "-o", exe, "-L", lib, "-lkidmp", "-lm",
                        "-Wl,-rpath," + lib], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    ncol, nsteps, nz = 2500, 3, 120                          # two pipeline chunks... of a batch the C side fills
    out = subprocess.run([exe, str(ncol), str(nsteps), "0,0"], capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr
    got = refused = multi = None
    for line in out.stdout.splitlines():
        if line.startswith("CAPI"):
            got = np.array([float(x) for x in line.split()[1:]])
        if line.startswith("REFUSED"):
            refused = line
        if line.startswith("MULTI"):
            multi = line.split()[1:]
    assert got is not None and refused is not None and multi is not None, out.stdout
    # the C host's multi-device leg (device list 0,0): two contexts, no element differs from the single-device run,
    # the RCCL-reduced rain sum equals the sum of the per-column values, nothing but rain reaches the ground
    assert int(multi[0]) == 2 and int(multi[1]) == 0, multi
    assert abs(float(multi[2]) - float(multi[6])) <= 1e-13 * abs(float(multi[6])) and [float(x) for x in multi[3:6]] == [0.0, 0.0, 0.0]
    assert int(refused.split()[1]) < 0 and "mixed-phase context needs" in refused

    n = ncol * nz
    raw_in = np.fromfile(str(tmp_path / "capi_in.bin"), dtype=np.float64)
    raw_out = np.fromfile(str(tmp_path / "capi_out.bin"), dtype=np.float64)
    assert raw_in.size == 7 * n and raw_out.size == 5 * n + 4 * ncol
    names_in, names_out = ("qv", "qc", "qr", "nr", "t", "p", "dz"), ("qv", "qc", "qr", "nr", "t")
    st = {k: raw_in[i * n:(i + 1) * n].reshape(ncol, nz).copy() for i, k in enumerate(names_in)}
    assert st["qc"].max() == 8.0e-4 and st["dz"].min() == 25.0 and st["qr"][-1].max() > st["qr"][0].max()
    ppt = np.zeros((ncol, 4))
    for _ in range(nsteps):
        gpu_warm.batch_step_host(st, 10.0, ppt=ppt)
    for i, k in enumerate(names_out):
        assert np.array_equal(st[k].ravel(), raw_out[i * n:(i + 1) * n]), k
    assert np.array_equal(ppt.ravel(), raw_out[5 * n:])
    assert not np.array_equal(st["qr"].ravel(), raw_in[2 * n:3 * n])       # the step did something
    assert abs(got[2] / st["qr"].sum() - 1) < 1e-11                       # and the printed sums are of that state
