"""The reference's run_data/*.data list-directed table cache format (M:3721-3727, M:3823-3828): our writer must
be readable by a Fortran `read(u,*)`, our parser must read what a Fortran `write(u,*)` produces.  The Fortran
side here is a tiny program of our own compiled with flang (the Fortran runtime is the format's authority)."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from kid_amd import KidmpError, cache_read_file, cache_write_file

FORTRAN = r"""
program cache_io
  implicit none
  double precision :: a(4,3,2), b(4,3,2)
  integer :: i, j, k
  character(256) :: mode, path
  call get_command_argument(1, mode)
  call get_command_argument(2, path)
  if (trim(mode) == 'write') then
     do k = 1, 2
        do j = 1, 3
           do i = 1, 4
              a(i,j,k) = 1.0d0/3.0d0 * i + 1.0d-7 * j - 2.5d3 * k
              b(i,j,k) = 0.0d0
           end do
        end do
     end do
     b(2,2,2) = 6.02214076d23; b(1,1,1) = -1.0d-300
     open(12, file=trim(path))
     write(12,*) a
     write(12,*) b
     close(12)
  else
     open(12, file=trim(path))
     read(12,*) a
     read(12,*) b
     close(12)
     write(*,'(es25.17e3)') a, b
  end if
end program
"""


@pytest.fixture(scope="module")
def fexe(tmp_path_factory):
    if shutil.which("flang") is None:
        pytest.skip("flang not available")
    d = tmp_path_factory.mktemp("fcache")
    src = d / "cache_io.f90"
    src.write_text(FORTRAN)
    exe = d / "cache_io"
    subprocess.check_call(["flang", "-O1", str(src), "-o", str(exe)])
    return str(exe)


def _expected():
    a = np.empty((4, 3, 2))
    for k in range(1, 3):
        for j in range(1, 4):
            for i in range(1, 5):
                a[i - 1, j - 1, k - 1] = 1.0 / 3.0 * i + 1.0e-7 * j - 2.5e3 * k
    b = np.zeros((4, 3, 2))
    b[1, 1, 1] = 6.02214076e23
    b[0, 0, 0] = -1.0e-300
    return a, b


def test_parser_reads_fortran_list_directed_output(fexe, tmp_path):
    path = str(tmp_path / "t.data")
    subprocess.check_call([fexe, "write", path])
    got = cache_read_file(path, 2, 24)
    a, b = _expected()
    np.testing.assert_array_equal(got[0], a.ravel(order="F"))      # flang writes round-trippable digits
    np.testing.assert_array_equal(got[1], b.ravel(order="F"))      # incl. any r*c repeat forms for the zeros


def test_fortran_reads_our_writer_output_exactly(fexe, tmp_path):
    path = str(tmp_path / "ours.data")
    a, b = _expected()
    cache_write_file(path, [a, b])
    out = subprocess.run([fexe, "read", path], capture_output=True, text=True, check=True).stdout.split()
    vals = np.array([float(x) for x in out])
    np.testing.assert_array_equal(vals[:24], a.ravel(order="F"))
    np.testing.assert_array_equal(vals[24:], b.ravel(order="F"))


def test_roundtrip_and_fortran_spellings(tmp_path):
    rng = np.random.default_rng(3)
    tabs = [rng.standard_normal((5, 4)) * 10.0 ** rng.integers(-30, 30, (5, 4)) for _ in range(3)]
    p = str(tmp_path / "rt.data")
    cache_write_file(p, tabs)
    back = cache_read_file(p, 3, 20)
    for t, g in zip(tabs, back):
        np.testing.assert_array_equal(g, t.ravel(order="F"))
    q = tmp_path / "spell.data"
    q.write_text(" 3*0.5, 1.25D-03 2.5d+2\n -1.0E0 ,7.  2*1.5-310\n")      # repeats, D exponents, commas, letterless exponent
    np.testing.assert_array_equal(cache_read_file(str(q), 1, 9)[0],
                                  [0.5, 0.5, 0.5, 1.25e-3, 250.0, -1.0, 7.0, 1.5e-310, 1.5e-310])
    with pytest.raises(KidmpError):
        cache_read_file(str(q), 1, 10)                                        # too few values
    with pytest.raises(KidmpError):
        cache_read_file(str(tmp_path / "missing.data"), 1, 1)
