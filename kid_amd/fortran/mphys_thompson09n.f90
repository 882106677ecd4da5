!
! mphys_thompson09n -- KiD-facing adapter of the MI355X Thompson-09n build.
!
! Drop-in for the reference adapter of the same name (/root/reference/mphys_thompson09n.f90, "W:"): same module
! name, same public entry `mphys_thompson09_interfacen` without arguments, data exchanged through KiD's own
! modules (parameters, column_variables, physconst, namelists, diagnostics; W:11-17), same results:
!   * d*_mphys tendencies = (state after microphysics - state)/dt - forcing          (W:198-245)
!   * surface precipitation diagnostics through save_dg, in the reference's order     (W:155-182, W:248-303)
!
! Design differences (see INTEGRATION.md):
!   * the reference loops `do i=1,nx` over one-column mp_thompson calls (W:54-246); here the nx columns are
!     packed once into a (nz, nx, slot) work array -- k fastest, KiD's own order, which is also the layout of the
!     C ABI -- and advanced by ONE batched call (one GPU launch);
!   * species are handled by a small slot table instead of one hand-written statement per species;
!   * the inputs the reference passes unset (nc1d, nwfa1d, nifa1d, w1d; W:36) get the scheme's non-aerosol
!     defaults (module_mp_thompson09n.f90 of the reference, lines 958-964).
!
module mphys_thompson09n

  Use parameters, only : num_h_moments, num_h_bins, nspecies, nz, dt &
       , h_names, mom_units, max_char_len, nx
  Use column_variables
  Use physconst, only : p0, r_on_cp, pi
  Use namelists, only : iiwarm, set_Nc
  Use module_mp_thompson09n
  Use diagnostics, only: save_dg, i_dgtime

  Implicit None

  ! public module variables of the reference adapter (W:22-24)
  logical :: micro_unset=.True.
  integer:: ih, imom
  character(max_char_len) :: name, units

  ! slots of the packed state, in the argument order of mp_thompson / the C ABI
  integer, parameter, private :: S_QV=1, S_QC=2, S_QI=3, S_QR=4, S_QS=5, S_QG=6, S_NI=7, S_NR=8, &
       S_NC=9, S_NWFA=10, S_NIFA=11, S_T=12, NSLOT=12
  ! prognostic hydrometeor moments KiD carries for this scheme: (state slot, KiD species, KiD moment)
  ! KiD species: 1 cloud, 2 rain, 3 ice, 4 snow, 5 graupel; moment 1 mass, 2 number (W:66-93)
  integer, parameter, private :: NHYD = 7
  integer, parameter, private :: hyd_slot(NHYD) = (/ S_QC, S_QR, S_NR, S_QI, S_NI, S_QS, S_QG /)
  integer, parameter, private :: hyd_spec(NHYD) = (/ 1,    2,    2,    3,    3,    4,    5    /)
  integer, parameter, private :: hyd_mom (NHYD) = (/ 1,    1,    2,    1,    2,    1,    1    /)
  ! precipitation diagnostics in the reference's call order rain, ice, snow, graupel (W:158-177):
  ! KiD species index and row of ppt(4,:) = (rain, snow, graupel, ice)
  integer, parameter, private :: dg_spec(4) = (/ 2, 3, 4, 5 /)
  integer, parameter, private :: dg_row (4) = (/ 1, 4, 2, 3 /)

contains

  Subroutine mphys_thompson09_interfacen

    ! Work arrays: the library's page-locked staging arrays themselves where KiD's REAL is what the selected arithmetic
    ! stores (mp_thompson_staging; no copy between here and PCIe), else heap arrays kept between calls (as automatic
    ! arrays they overflow the stack from nx ~ 500 on).  pres, wvel, dzc are fo(:,:,1:3).
    real, pointer :: st(:,:,:), fo(:,:,:), ppt(:,:)
    real, allocatable, target, save :: st_own(:,:,:), fo_own(:,:,:), ppt_own(:,:)
    real, allocatable, save :: total(:)
    real, allocatable, save :: pptrain_2d_prof(:,:)          ! W:32; saved as 'total_ppt_level' for nx > 1 (W:304-307)
    real :: rho
    logical :: staged
    integer :: i, k, m, s

    if (micro_unset) then                        ! W:100-103 (ahead of the gather: the staging arrays belong to the library)
       call thompson_init
       micro_unset = .False.
    end if
    call mp_thompson_staging(nx, nz, st, fo, ppt, staged)
    if (.not. staged) then
       if (allocated(st_own)) then
          if (size(st_own,1) /= nz .or. size(st_own,2) /= nx) deallocate(st_own, fo_own, ppt_own)
       end if
       if (.not. allocated(st_own)) allocate(st_own(nz,nx,NSLOT), fo_own(nz,nx,3), ppt_own(4,nx))
       st => st_own;  fo => fo_own;  ppt => ppt_own
    end if
    if (allocated(total)) then
       if (size(total) /= nx) deallocate(total)
    end if
    if (.not. allocated(total)) allocate(total(nx))

    ! ---- gather: state + (advective + divergence forcing)*dt, W:59-97 ----
    ! Columns are independent: the loop over i is shared among OpenMP threads when the model is built with OpenMP
    ! (the directives are comments otherwise).  At nx = 10^4 this host-side packing, not the GPU, bounds a KiD step.
    !$omp parallel do default(shared) private(i, k, m, s) schedule(static) if(nx >= 256)
    do i = 1, nx
       st(:,i,S_T)  = (theta(:,i) + (dtheta_adv(:,i) + dtheta_div(:,i))*dt)*exner(:,i)
       fo(:,i,1)    = p0*exner(:,i)**(1./r_on_cp)
       fo(:,i,3)    = dz(:)
       st(:,i,S_QV) = qv(:,i) + (dqv_adv(:,i) + dqv_div(:,i))*dt
       do m = 1, NHYD
          if (iiwarm .and. hyd_spec(m) > 2) cycle
          s = hyd_slot(m)
          do k = 1, nz
             st(k,i,s) = hydrometeors(k,i,hyd_spec(m))%moments(1,hyd_mom(m)) &
                  + (dhydrometeors_adv(k,i,hyd_spec(m))%moments(1,hyd_mom(m)) &
                  +  dhydrometeors_div(k,i,hyd_spec(m))%moments(1,hyd_mom(m)))*dt
          end do
       end do
    end do
    !$omp end parallel do
    ! What the reference leaves unset (nc1d, nwfa1d, nifa1d, w1d; W:36): with is_aerosol_aware they are read, and get
    ! the scheme's non-aerosol defaults (M:958-964) and no updraft here; without it they are left out of the call and
    ! the library forms the same defaults on the GPU, so they never cross PCIe.
    if (is_aerosol_aware) then
       fo(:,:,2) = 0.0
       if (iiwarm) then
          ! a warm run keeps qc1d..qg1d at their initial zeros (W:46-52): the aerosol-aware call below passes all
          ! twelve slots, and the staging memory is neither zeroed by the library nor by ALLOCATE
          st(:,:,S_QI) = 0.0;  st(:,:,S_QS) = 0.0;  st(:,:,S_QG) = 0.0;  st(:,:,S_NI) = 0.0
       end if
       do i = 1, nx
          do k = 1, nz
             rho = 0.622*fo(k,i,1)/(287.04*st(k,i,S_T)*(st(k,i,S_QV)+0.622))
             st(k,i,S_NC)   = set_Nc*1.e6/rho
             st(k,i,S_NWFA) = 11.1E6/rho
             st(k,i,S_NIFA) = 0.5E6*0.01/rho
          end do
       end do
    end if

    ! ---- all nx columns in one call (replaces the loop around W:143-152) ----
    ppt = 0.0
    if (is_aerosol_aware) then
       call mp_thompson_batch(nx, nz, dt, st(:,:,S_QV), st(:,:,S_QC), st(:,:,S_QI), st(:,:,S_QR), st(:,:,S_QS), &
            st(:,:,S_QG), st(:,:,S_NI), st(:,:,S_NR), st(:,:,S_NC), st(:,:,S_NWFA), st(:,:,S_NIFA), st(:,:,S_T), &
            fo(:,:,1), fo(:,:,2), fo(:,:,3), ppt)
    else if (iiwarm) then                        ! a warm run: the frozen species stay zero (W:46-52) and stay at home
       call mp_thompson_batch(nx, nz, dt, qv=st(:,:,S_QV), qc=st(:,:,S_QC), qr=st(:,:,S_QR), nr=st(:,:,S_NR), &
            t=st(:,:,S_T), p=fo(:,:,1), dz=fo(:,:,3), ppt=ppt)
    else
       call mp_thompson_batch(nx, nz, dt, qv=st(:,:,S_QV), qc=st(:,:,S_QC), qi=st(:,:,S_QI), qr=st(:,:,S_QR), &
            qs=st(:,:,S_QS), qg=st(:,:,S_QG), ni=st(:,:,S_NI), nr=st(:,:,S_NR), t=st(:,:,S_T), p=fo(:,:,1), dz=fo(:,:,3), ppt=ppt)
    end if

    ! ---- back out the microphysics tendencies, W:198-245 ----
    !$omp parallel do default(shared) private(i, k, m, s) schedule(static) if(nx >= 256)
    do i = 1, nx
       dtheta_mphys(:,i) = (st(:,i,S_T)/exner(:,i) - theta(:,i))/dt - (dtheta_adv(:,i) + dtheta_div(:,i))
       dqv_mphys(:,i)    = (st(:,i,S_QV) - qv(:,i))/dt - (dqv_adv(:,i) + dqv_div(:,i))
       do m = 1, NHYD
          if (iiwarm .and. hyd_spec(m) > 2) cycle
          s = hyd_slot(m)
          do k = 1, nz
             dhydrometeors_mphys(k,i,hyd_spec(m))%moments(1,hyd_mom(m)) = &
                  (st(k,i,s) - hydrometeors(k,i,hyd_spec(m))%moments(1,hyd_mom(m)))/dt &
                  - (dhydrometeors_adv(k,i,hyd_spec(m))%moments(1,hyd_mom(m)) &
                  +  dhydrometeors_div(k,i,hyd_spec(m))%moments(1,hyd_mom(m)))
          end do
       end do
    end do
    !$omp end parallel do

    ! ---- surface precipitation diagnostics ----
    imom = 1
    units = trim(mom_units(imom))//' m'
    total = ppt(4,:) + ppt(1,:) + ppt(2,:) + ppt(3,:)          ! ice + rain + snow + graupel, as W:181
    if (nx == 1) then                                          ! W:155-182
       do m = 1, 4
          ih = dg_spec(m)
          name = 'surface_ppt_for_'//trim(h_names(ih))
          call save_dg(ppt(dg_row(m),1), name, i_dgtime, units, dim='time')
       end do
       name = 'total_surface_ppt'
       call save_dg(total(1)/nx, name, i_dgtime, units, dim='time')
    else                                                       ! W:248-303: domain means, then every column
       do m = 1, 4
          ih = dg_spec(m)
          name = 'surface_ppt_for_'//trim(h_names(ih))
          call save_dg(ppt(dg_row(m),:)/nx, name, i_dgtime, units, dim='time')
       end do
       name = 'total_surface_ppt'
       call save_dg(total/nx, name, i_dgtime, units, dim='time')
       do m = 1, 4
          ih = dg_spec(m)
          name = 'surface_ppt_for_'//trim(h_names(ih))
          call save_dg(ppt(dg_row(m),:), name, i_dgtime, units, dim='time')
       end do
       name = 'total_surface_ppt'
       call save_dg(total, name, i_dgtime, units, dim='time')
       ! save precip flux at all levels and columns, W:304-307.  The reference saves pptrain_2d_prof(nz,nx), an array it
       ! never assigns (W:191 is commented out), i.e. undefined values under a name every stock KiD output carries.
       ! Defined semantics here (U6): the name, units, dim='z,x' and shape of the reference, all values zero.
       if (.not. allocated(pptrain_2d_prof)) then
          allocate(pptrain_2d_prof(nz, nx))
          pptrain_2d_prof = 0.0
       end if
       name = 'total_ppt_level'
       call save_dg(pptrain_2d_prof, name, i_dgtime, units, dim='z,x')
    end if

  end Subroutine mphys_thompson09_interfacen

end module mphys_thompson09n
