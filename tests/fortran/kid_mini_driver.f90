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
!   further arguments, in any order after the sixth (use `-` for skipped positional ones):
!     forcing=1    non-zero dtheta_adv/div, dqv_adv/div, dhydrometeors_adv/div: a prescribed updraft (advective
!                  tendencies -w d/dz of every field) plus a weak divergence term, the SURVEY 8d config-1 variant; the
!                  time loop then adds adv + div + mphys, as KiD does
!     mphys=<n>    write inputs, forcing terms and the d*_mphys outputs of step n to mphys_dump.txt
!     aero=1       module_mp_thompson09n's is_aerosol_aware = .true.
!     reuse=1      switches' l_reuse_thompson_lookup = .true. (run_data/*.data are read when they exist)
!     devices=a,b  spread the columns over these HIP devices (kidmp_ndevices / kidmp_devices)
!     time=1       print the wall-clock time of the step loop (system_clock around it) as
!                  "TIME <steps> <seconds> <column-steps/s>"
! Prints the end-state sums of column 1 and of column nx.
program kid_mini_driver
  use parameters, only: nz, nx, dt
  use column_variables
  use namelists, only: iiwarm, set_Nc
  use switches, only: l_reuse_thompson_lookup
  use diagnostics, only: recording, nlog, dump_log
  use mphys_thompson09n, only: mphys_thompson09_interfacen
  use module_mp_thompson09n, only: thompson_finalize, kidmp_arith, l_rate_diagnostics, is_aerosol_aware, &
       kidmp_ndevices, kidmp_devices, kidmp_precip_sums, kidmp_precip_sums_valid
  implicit none
  integer :: k, i, n, j, nsteps, dump_step, mphys_step, a, eq, ios, ndv, pos, nxt, up, dn
  integer(8) :: c0, c1, crate
  real :: z, p, t, es, qsat, w
  logical :: forcing, timing
  character(64) :: arg, which, key, val

  nsteps = 360; which = 'warm'; dump_step = 0; mphys_step = 0; forcing = .false.; timing = .false.
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
  do a = 6, command_argument_count()
     call get_command_argument(a, arg)
     eq = index(arg, '=')
     if (eq == 0) cycle
     key = arg(1:eq-1);  val = arg(eq+1:)
     select case (trim(key))
     case ('forcing');  forcing = trim(val) == '1'
     case ('mphys');    read(val,*) mphys_step
     case ('aero');     is_aerosol_aware = trim(val) == '1'
     case ('reuse');    l_reuse_thompson_lookup = trim(val) == '1'
     case ('time');     timing = trim(val) == '1'
     case ('devices')
        ndv = 0;  pos = 1
        do while (pos <= len_trim(val) .and. ndv < 8)
           nxt = index(val(pos:), ',')
           if (nxt == 0) nxt = len_trim(val) - pos + 2
           ndv = ndv + 1
           read(val(pos:pos+nxt-2), *, iostat=ios) kidmp_devices(ndv)
           pos = pos + nxt
        end do
        kidmp_ndevices = ndv
     case default
        write(*,'(2a)') ' kid_mini_driver: unknown option ', trim(arg)
        stop 2
     end select
  end do
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
  if (forcing) then
     ! A prescribed updraft w(z) = 2 sin(pi z / ztop) m/s advects every field (tendency -w d/dz, centred differences,
     ! one-sided at the ends) and a weak divergence term removes 2e-5 of each field per second; columns differ by 5 %.
     do i = 1, nx
        do k = 1, nz
           z = (k-0.5)*dz(k)
           w = 2.0*sin(3.14159265*z/(nz*dz(k)))*(1. + 0.05*(i-1))
           up = min(k+1, nz);  dn = max(k-1, 1)
           dtheta_adv(k,i) = -w*(theta(up,i) - theta(dn,i))/((up-dn)*dz(k))
           dqv_adv(k,i)    = -w*(qv(up,i) - qv(dn,i))/((up-dn)*dz(k))
           dtheta_div(k,i) = -2.e-5*(theta(k,i) - theta(1,i))
           dqv_div(k,i)    = -2.e-5*qv(k,i)
           do j = 1, 5
              dhydrometeors_adv(k,i,j)%moments = -w*(hydrometeors(up,i,j)%moments - hydrometeors(dn,i,j)%moments) &
                   /((up-dn)*dz(k))
              dhydrometeors_div(k,i,j)%moments = -2.e-5*hydrometeors(k,i,j)%moments
           end do
        end do
     end do
  end if
  if (dump_step > 0) then                    ! the inputs of the dumped step, for the oracle side of the test
     open(22, file='dg_inputs.txt', status='replace')
  end if
  if (timing) then                            ! initialisation (tables, staging memory, first touch) is not the step loop
     call mphys_thompson09_interfacen
     call system_clock(c0, crate)
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
     if (n == mphys_step) then               ! everything the adapter read and everything it produced, for the oracle
        open(23, file='mphys_dump.txt', status='replace')
        do i = 1, nx
           do k = 1, nz
              write(23,'(38es25.17)') theta(k,i), exner(k,i), qv(k,i), dz(k), &
                   hydrometeors(k,i,1)%moments(1,1), hydrometeors(k,i,2)%moments(1,1), hydrometeors(k,i,2)%moments(1,2), &
                   hydrometeors(k,i,3)%moments(1,1), hydrometeors(k,i,3)%moments(1,2), hydrometeors(k,i,4)%moments(1,1), &
                   hydrometeors(k,i,5)%moments(1,1), &
                   dtheta_adv(k,i), dtheta_div(k,i), dqv_adv(k,i), dqv_div(k,i), &
                   dhydrometeors_adv(k,i,1)%moments(1,1), dhydrometeors_adv(k,i,2)%moments(1,1), dhydrometeors_adv(k,i,2)%moments(1,2), &
                   dhydrometeors_adv(k,i,3)%moments(1,1), dhydrometeors_adv(k,i,3)%moments(1,2), dhydrometeors_adv(k,i,4)%moments(1,1), &
                   dhydrometeors_adv(k,i,5)%moments(1,1), &
                   dhydrometeors_div(k,i,1)%moments(1,1), dhydrometeors_div(k,i,2)%moments(1,1), dhydrometeors_div(k,i,2)%moments(1,2), &
                   dhydrometeors_div(k,i,3)%moments(1,1), dhydrometeors_div(k,i,3)%moments(1,2), dhydrometeors_div(k,i,4)%moments(1,1), &
                   dhydrometeors_div(k,i,5)%moments(1,1), &
                   dtheta_mphys(k,i), dqv_mphys(k,i), &
                   dhydrometeors_mphys(k,i,1)%moments(1,1), dhydrometeors_mphys(k,i,2)%moments(1,1), dhydrometeors_mphys(k,i,2)%moments(1,2), &
                   dhydrometeors_mphys(k,i,3)%moments(1,1), dhydrometeors_mphys(k,i,3)%moments(1,2), dhydrometeors_mphys(k,i,4)%moments(1,1), &
                   dhydrometeors_mphys(k,i,5)%moments(1,1)
           end do
        end do
        close(23)
     end if
     if (timing) cycle                        ! timing runs repeat the same step: no state update on the host clock
     ! KiD's time loop: every tendency is added (the forcing terms are zero unless forcing=1)
     theta = theta + dt*(dtheta_mphys + dtheta_adv + dtheta_div)
     qv = qv + dt*(dqv_mphys + dqv_adv + dqv_div)
     do j = 1, 5
        do i = 1, nx
           do k = 1, nz
              hydrometeors(k,i,j)%moments = hydrometeors(k,i,j)%moments + dt*(dhydrometeors_mphys(k,i,j)%moments &
                   + dhydrometeors_adv(k,i,j)%moments + dhydrometeors_div(k,i,j)%moments)
           end do
        end do
     end do
  end do
  if (timing) then
     call system_clock(c1)
     write(*,'(a,i0,1x,es14.6,1x,es14.6)') 'TIME ', nsteps, real(c1-c0,8)/real(crate,8), &
          real(nx,8)*real(nsteps,8)*real(crate,8)/real(max(c1-c0,1_8),8)
  end if
  if (kidmp_precip_sums_valid()) write(*,'(a,4es24.16)') 'PSUM ', kidmp_precip_sums
  write(*,'(a,4es24.16)') 'KATB ', sum(qv(:,1)), sum(hydrometeors(:,1,1)%moments(1,1)), &
       sum(hydrometeors(:,1,2)%moments(1,1)), sum(hydrometeors(:,1,2)%moments(1,2))
  write(*,'(a,4es24.16)') 'KATBN', sum(qv(:,nx)), sum(hydrometeors(:,nx,1)%moments(1,1)), &
       sum(hydrometeors(:,nx,2)%moments(1,1)), sum(hydrometeors(:,nx,2)%moments(1,2))
  call thompson_finalize
end program kid_mini_driver
