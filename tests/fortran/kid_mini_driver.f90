! BASELINE config 1 through the Fortran boundary, driven exactly the way the KiD time loop drives a microphysics
! scheme -- call mphys_thompson09_interfacen, then state += dt * d(state)_mphys.
!
!   kid_mini_driver [nx [nsteps [case [dump_step [arith]]]]]
!     nx         replicated columns (default 1)
!     nsteps     time steps (default 360)
!     case       warm  = the KiD 1-D warm-rain case (KAT-B of SURVEY 9h: nz=120, dz=25 m, dt=10 s, zero forcing)
!                mixed = the mixed-phase deep-convection sounding (KAT-A of SURVEY 9h, dz=125 m) fed through the wrapper
!                dry   = the warm case without hydrometeors at 30 % of its vapour: mp_thompson returns at no_micro
!     arith      p64 (default) | p32n | f32: module_mp_thompson09n's kidmp_arith (the last two need 4-byte default REAL)
!     dump_step  write every save_dg call made during that step to dg_dump.txt (the recording `diagnostics` stub)
!     norates    (6th argument) module_mp_thompson09n's l_rate_diagnostics = .false.
! Prints the end-state sums of column 1 and of column nx.
program kid_mini_driver
  use parameters, only: nz, nx, dt
  use column_variables
  use namelists, only: iiwarm, set_Nc
  use diagnostics, only: recording, nlog, dump_log
  use mphys_thompson09n, only: mphys_thompson09_interfacen
  use module_mp_thompson09n, only: thompson_finalize, kidmp_arith, l_rate_diagnostics
  implicit none
  integer :: k, i, n, j, nsteps, dump_step
  real :: z, p, t, es, qsat
  character(32) :: arg, which

  nsteps = 360; which = 'warm'; dump_step = 0
  if (command_argument_count() >= 1) then
     call get_command_argument(1, arg); read(arg,*) nx
  end if
  if (command_argument_count() >= 2) then
     call get_command_argument(2, arg); read(arg,*) nsteps
  end if
  if (command_argument_count() >= 3) call get_command_argument(3, which)
  if (command_argument_count() >= 4) then
     call get_command_argument(4, arg); read(arg,*) dump_step
  end if
  if (command_argument_count() >= 5) call get_command_argument(5, kidmp_arith)   ! p64 (default) | p32n | f32
  if (command_argument_count() >= 6) then
     call get_command_argument(6, arg)
     if (trim(arg) == 'norates') l_rate_diagnostics = .false.      ! timing runs: no replay of the 36 rate diagnostics
  end if
  iiwarm = trim(which) /= 'mixed'; set_Nc = 100.0
  call alloc_columns(nz, nx)
  do i = 1, nx
     do k = 1, nz
        if (iiwarm) then
           z = (k-0.5)*25.
           dz(k) = 25.
           p = 1.e5*(1.-2.2557e-5*z)**5.2559
           exner(k,i) = (p/1.e5)**(287.058/1005.)
           t = 297. - 6.5e-3*z
           theta(k,i) = t/exner(k,i)
           qv(k,i) = 0.015 - 0.004*z/3000.
           if (trim(which) == 'dry') then          ! nothing to do: every call leaves through no_micro (M:1540)
              qv(k,i) = 0.3*qv(k,i)
           else if (z > 800. .and. z < 2000.) then
              hydrometeors(k,i,1)%moments(1,1) = 8.e-4
              hydrometeors(k,i,2)%moments(1,1) = 3.e-4
              hydrometeors(k,i,2)%moments(1,2) = 2.e4
           end if
        else
           z = (k-0.5)*125.
           dz(k) = 125.
           p = 1.e5*(1.-2.2557e-5*z)**5.2559
           exner(k,i) = (p/1.e5)**(287.058/1005.)
           t = max(210., 300. - 6.5e-3*z)
           theta(k,i) = t/exner(k,i)
           es = 611.2*exp(17.67*(t-273.15)/(t-29.65))
           qsat = 0.622*es/(p-es)
           qv(k,i) = 0.7*qsat
           if (z > 1000. .and. z < 4000.) then
              qv(k,i) = 1.02*qsat
              hydrometeors(k,i,1)%moments(1,1) = 1.e-3
              hydrometeors(k,i,2)%moments(1,1) = 5.e-4
              hydrometeors(k,i,2)%moments(1,2) = 5.e3
           else if (z > 4000. .and. z < 11000.) then
              qv(k,i) = qsat
              hydrometeors(k,i,1)%moments(1,1) = 2.e-4
              hydrometeors(k,i,2)%moments(1,1) = 1.e-4
              hydrometeors(k,i,2)%moments(1,2) = 1.e3
              hydrometeors(k,i,3)%moments(1,1) = 1.e-4
              hydrometeors(k,i,3)%moments(1,2) = 1.e5
              hydrometeors(k,i,4)%moments(1,1) = 1.e-3
              hydrometeors(k,i,5)%moments(1,1) = 2.e-3
           end if
           ! columns differ a little so that the per-column call form carries information
           hydrometeors(k,i,2)%moments(1,1) = hydrometeors(k,i,2)%moments(1,1)*(1. + 0.1*(i-1))
        end if
     end do
  end do
  if (dump_step > 0) then                    ! the inputs of the dumped step, for the oracle side of the test
     open(22, file='dg_inputs.txt', status='replace')
  end if
  do n = 1, nsteps
     recording = n == dump_step
     if (recording) then
        nlog = 0
        do i = 1, nx
           do k = 1, nz
              write(22,'(12es25.17)') theta(k,i), exner(k,i), qv(k,i), dz(k), &
                   hydrometeors(k,i,1)%moments(1,1), hydrometeors(k,i,2)%moments(1,1), hydrometeors(k,i,2)%moments(1,2), &
                   hydrometeors(k,i,3)%moments(1,1), hydrometeors(k,i,3)%moments(1,2), hydrometeors(k,i,4)%moments(1,1), &
                   hydrometeors(k,i,5)%moments(1,1), 0.
           end do
        end do
        close(22)
     end if
     call mphys_thompson09_interfacen
     if (recording) call dump_log('dg_dump.txt')
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
