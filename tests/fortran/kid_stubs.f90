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
  integer :: i_dgtime = 1
  real :: last_scalar(8) = 0.
  integer :: n_scalar = 0
  interface save_dg
     module procedure save_dg_scalar, save_dg_1d
  end interface
contains
  subroutine save_dg_scalar(v, name, it, units, dim)
    real, intent(in) :: v
    character(*), intent(in) :: name, units, dim
    integer, intent(in) :: it
    n_scalar = mod(n_scalar, 8) + 1
    last_scalar(n_scalar) = v
    if (.false.) print *, name, it, units, dim
  end subroutine save_dg_scalar
  subroutine save_dg_1d(v, name, it, units, dim)
    real, intent(in) :: v(:)
    character(*), intent(in) :: name, units, dim
    integer, intent(in) :: it
    if (.false.) print *, v(1), name, it, units, dim
  end subroutine save_dg_1d
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
