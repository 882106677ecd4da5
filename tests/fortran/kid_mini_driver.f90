! BASELINE config 1 through the Fortran boundary: the KiD 1-D warm-rain case (KAT-B of SURVEY 9h:
! nz=120, dz=25 m, dt=10 s, 360 steps, zero forcing) driven exactly the way the KiD time loop drives a
! microphysics scheme -- call mphys_thompson09_interfacen, then state += dt * d(state)_mphys.
! Optional argument: nx (replicated columns, default 1).  Prints the end-state sums of column 1.
program kid_mini_driver
  use parameters, only: nz, nx, dt
  use column_variables
  use namelists, only: iiwarm, set_Nc
  use mphys_thompson09n, only: mphys_thompson09_interfacen
  use module_mp_thompson09n, only: thompson_finalize
  implicit none
  integer :: k, i, n, j, nsteps
  real :: z, p, t
  character(32) :: arg

  nsteps = 360
  if (command_argument_count() >= 1) then
     call get_command_argument(1, arg); read(arg,*) nx
  end if
  if (command_argument_count() >= 2) then
     call get_command_argument(2, arg); read(arg,*) nsteps
  end if
  iiwarm = .true.; set_Nc = 100.0
  call alloc_columns(nz, nx)
  do i = 1, nx
     do k = 1, nz
        z = (k-0.5)*25.
        dz(k) = 25.
        p = 1.e5*(1.-2.2557e-5*z)**5.2559
        exner(k,i) = (p/1.e5)**(287.058/1005.)
        t = 297. - 6.5e-3*z
        theta(k,i) = t/exner(k,i)
        qv(k,i) = 0.015 - 0.004*z/3000.
        if (z > 800. .and. z < 2000.) then
           hydrometeors(k,i,1)%moments(1,1) = 8.e-4
           hydrometeors(k,i,2)%moments(1,1) = 3.e-4
           hydrometeors(k,i,2)%moments(1,2) = 2.e4
        end if
     end do
  end do
  do n = 1, nsteps
     call mphys_thompson09_interfacen
     theta = theta + dt*dtheta_mphys
     qv = qv + dt*dqv_mphys
     do j = 1, 5
        do i = 1, nx
           do k = 1, nz
              hydrometeors(k,i,j)%moments = hydrometeors(k,i,j)%moments + dt*dhydrometeors_mphys(k,i,j)%moments
           end do
        end do
     end do
  end do
  write(*,'(a,4es24.16)') 'KATB ', sum(qv(:,1)), sum(hydrometeors(:,1,1)%moments(1,1)), &
       sum(hydrometeors(:,1,2)%moments(1,1)), sum(hydrometeors(:,1,2)%moments(1,2))
  write(*,'(a,4es24.16)') 'KATBN', sum(qv(:,nx)), sum(hydrometeors(:,nx,1)%moments(1,1)), &
       sum(hydrometeors(:,nx,2)%moments(1,1)), sum(hydrometeors(:,nx,2)%moments(1,2))
  call thompson_finalize
end program kid_mini_driver
