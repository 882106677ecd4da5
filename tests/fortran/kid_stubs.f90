! Minimal stand-ins for the KiD driver modules that the build-owned drop-in
! modules (kid_amd/fortran/*.f90) USE.  TEST SCAFFOLDING for OUR OWN code only:
! the real KiD driver is not in the reference mount; nothing here is used to
! build or run the reference.
module typeKind
  integer, parameter :: wp = kind(1.0)
end module typeKind

module parameters
  integer, parameter :: max_char_len = 200
  integer, parameter :: nspecies = 5, num_h_moments(5) = (/1,2,2,1,1/), num_h_bins(5) = 1
  integer :: nz = 120, nx = 1
  real :: dt = 10.0
  character(10) :: h_names(5) = (/'cloud     ','rain      ','ice       ','snow      ','graupel   '/)
  character(10) :: mom_units(2) = (/'kg/kg     ','/kg       '/)
end module parameters

module switches
  logical :: l_sediment = .true., l_reuse_thompson_lookup = .false.
end module switches

module namelists
  logical :: iiwarm = .true.
  real :: set_Nc = 100.0
end module namelists

module physconst
  real :: p0 = 1.e5, r_on_cp = 287.058/1005., pi = 3.14159265358979
end module physconst

module diagnostics
  ! Records every save_dg call (form, name, indices, value) so that the tests can assert names, order and
  ! values of what the drop-in modules emit; KiD's real module writes them to netCDF.
  integer :: i_dgtime = 1
  integer, parameter :: maxlog = 200000
  integer :: nlog = 0
  logical :: recording = .false.
  character(40) :: log_name(maxlog)
  character(8) :: log_form(maxlog), log_units(maxlog), log_dim(maxlog)
  integer :: log_k(maxlog), log_i(maxlog)
  double precision :: log_v(maxlog)
  interface save_dg
     module procedure save_dg_scalar, save_dg_1d, save_dg_2d, save_dg_k_dp, save_dg_ki_dp
  end interface
contains
  subroutine put(form, name, k, i, v, units, dim)
    character(*), intent(in) :: form, name, units, dim
    integer, intent(in) :: k, i
    double precision, intent(in) :: v
    if (.not. recording) return
    if (nlog >= maxlog) stop 'diagnostics stub: log full'
    nlog = nlog + 1
    log_form(nlog) = form;  log_name(nlog) = name;  log_k(nlog) = k;  log_i(nlog) = i;  log_v(nlog) = v
    log_units(nlog) = units; log_dim(nlog) = dim
  end subroutine put
  subroutine save_dg_scalar(v, name, it, units, dim)            ! scalar form, W:162
    real, intent(in) :: v
    character(*), intent(in) :: name, units, dim
    integer, intent(in) :: it
    call put('scalar', name, 0, 0, dble(v), units, dim)
    if (.false.) print *, it
  end subroutine save_dg_scalar
  subroutine save_dg_1d(v, name, it, units, dim)                ! 1-D form, W:255
    real, intent(in) :: v(:)
    character(*), intent(in) :: name, units, dim
    integer, intent(in) :: it
    integer :: i
    do i = 1, size(v)
       call put('array', name, 0, i, dble(v(i)), units, dim)
    end do
    if (.false.) print *, it
  end subroutine save_dg_1d
  subroutine save_dg_2d(v, name, it, units, dim)                ! 2-D form, W:307 (dim='z,x')
    real, intent(in) :: v(:,:)
    character(*), intent(in) :: name, units, dim
    integer, intent(in) :: it
    integer :: i, k
    do i = 1, size(v, 2)
       do k = 1, size(v, 1)
          call put('2d', name, k, i, dble(v(k,i)), units, dim)
       end do
    end do
    if (.false.) print *, it
  end subroutine save_dg_2d
  subroutine save_dg_k_dp(k, v, name, it, units, dim)           ! per-level rate form, nx == 1 (M:2967)
    integer, intent(in) :: k, it
    double precision, intent(in) :: v
    character(*), intent(in) :: name, units, dim
    call put('k', name, k, 0, v, units, dim)
    if (.false.) print *, it
  end subroutine save_dg_k_dp
  subroutine save_dg_ki_dp(k, i, v, name, it, units, dim)       ! per-level rate form, nx > 1 (M:3044)
    integer, intent(in) :: k, i, it
    double precision, intent(in) :: v
    character(*), intent(in) :: name, units, dim
    call put('ki', name, k, i, v, units, dim)
    if (.false.) print *, it
  end subroutine save_dg_ki_dp
  subroutine dump_log(path)
    character(*), intent(in) :: path
    integer :: n
    open(21, file=path, status='replace')
    do n = 1, nlog
       write(21,'(a,1x,a,1x,i0,1x,i0,1x,es25.17,1x,a,1x,a)') trim(log_form(n)), trim(log_name(n)), log_k(n), log_i(n), &
            log_v(n), trim(log_units(n)), trim(log_dim(n))
    end do
    close(21)
  end subroutine dump_log
end module diagnostics

module column_variables
  type species
     real :: moments(1,2)
  end type species
  real, allocatable :: theta(:,:), dtheta_adv(:,:), dtheta_div(:,:), dtheta_mphys(:,:), exner(:,:), &
       qv(:,:), dqv_adv(:,:), dqv_div(:,:), dqv_mphys(:,:), dz(:)
  type(species), allocatable :: hydrometeors(:,:,:), dhydrometeors_adv(:,:,:), dhydrometeors_div(:,:,:), &
       dhydrometeors_mphys(:,:,:)
contains
  subroutine alloc_columns(nz, nx)
    integer, intent(in) :: nz, nx
    integer :: i, j, k
    allocate(theta(nz,nx), dtheta_adv(nz,nx), dtheta_div(nz,nx), dtheta_mphys(nz,nx), exner(nz,nx), &
         qv(nz,nx), dqv_adv(nz,nx), dqv_div(nz,nx), dqv_mphys(nz,nx), dz(nz))
    allocate(hydrometeors(nz,nx,5), dhydrometeors_adv(nz,nx,5), dhydrometeors_div(nz,nx,5), &
         dhydrometeors_mphys(nz,nx,5))
    dtheta_adv = 0.; dtheta_div = 0.; dtheta_mphys = 0.; dqv_adv = 0.; dqv_div = 0.; dqv_mphys = 0.
    do j = 1, 5
       do i = 1, nx
          do k = 1, nz
             hydrometeors(k,i,j)%moments = 0.
             dhydrometeors_adv(k,i,j)%moments = 0.
             dhydrometeors_div(k,i,j)%moments = 0.
             dhydrometeors_mphys(k,i,j)%moments = 0.
          end do
       end do
    end do
  end subroutine alloc_columns
end module column_variables
