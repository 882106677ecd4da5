#!/bin/bash
# Which math functions does a flang-built default-REAL program call?  ("P32n and the native digits": the
# oracle's P32n mode takes powf / expf / logf / log10f from glibc through <tgmath.h>; this shows that flang lowers the Fortran
# generic intrinsics and REAL**REAL to the same functions and links no other math library.)  Our own ten lines of Fortran, not
# the reference.   usage: tools/flang_intrinsics_probe.sh
set -e
d=$(mktemp -d); trap 'rm -rf $d' EXIT
cat > $d/p.f90 <<'F90'
subroutine probe(n, x, y, o)
  integer, intent(in) :: n
  real, intent(in) :: x(n), y(n)
  real, intent(out) :: o(n,8)
  double precision :: d
  integer :: i
  do i = 1, n
    o(i,1) = x(i)**y(i);   o(i,2) = x(i)**3;      o(i,3) = 10.**y(i);   o(i,4) = exp(y(i))
    o(i,5) = alog(x(i));   o(i,6) = alog10(x(i)); o(i,7) = sqrt(x(i))
    d = dble(x(i));        o(i,8) = d**y(i)
  end do
end subroutine
program t
  real :: x(1), y(1), o(1,8)
  read(*,*) x(1), y(1)
  call probe(1, x, y, o)
  print '(8(Z8.8,1X))', transfer(o(1,:), (/1,1,1,1,1,1,1,1/))
end program
F90
flang -O2 -c $d/p.f90 -o $d/p.o
echo "undefined math symbols of the object:"; nm $d/p.o | awk '$1=="U" && $2 !~ /^_Fortran/ {print "  " $2}'
flang -O2 $d/p.f90 -o $d/p
echo "shared libraries of the program:"; ldd $d/p | sed 's/^/  /'
echo "1.2345 2.5" | $d/p
