!
! module_mp_thompson09n -- drop-in replacement for the reference module of the
! same name (/root/reference/module_mp_thompson09n.f90, "M:").  It keeps the two
! procedures the KiD adapter calls, with the reference's dummy lists:
!
!     thompson_init()                                   M:374
!     mp_thompson(qv1d, ..., kts, kte, dt, ii, jj)      M:1156-1177
!
! and adds the batched form of the adapter's `do i=1,nx` loop (W:54-246):
!
!     mp_thompson_batch(ncol, nz, dt, qv, ..., ppt)
!
! Bodies are ISO_C_BINDING calls into libkidmp.so (include/kidmp.h), where the
! column physics runs as hand-written HIP kernels on an MI355X.  Arithmetic (kidmp_arith):
!   'p64'   (default) state arrays are copied to REAL(c_double) temporaries and the step runs in binary64 -- the
!           parity build; works whether KiD is compiled with 4-byte or 8-byte default REAL;
!   'p32n'  KiD's default 4-byte REAL only: the REAL arrays go to the GPU as they are and the kernel keeps the
!           reference's own split -- what it declares REAL in binary32, its DOUBLE PRECISION rates in binary64
!           (the reference as shipped);
!   'f32'   likewise, everything binary32.
!
! Like the reference it takes its switches from KiD's own modules:
! iiwarm, set_Nc (namelists, M:22), l_sediment and l_reuse_thompson_lookup (switches, M:20) and nx (parameters, M:23).
! l_reuse_thompson_lookup keeps the reference's meaning (M:3717-3729, M:3864-3895): a run_data/racg_thompson09.data /
! run_data/racs_thompson09.data that exists is READ when the switch is set; otherwise the tables are built (on the GPU)
! and WRITTEN there, in the reference's list-directed format (without a run_data directory nothing is written; the
! reference aborts in that case, M:3718).
!
! More than one GPU: kidmp_ndevices > 1 with kidmp_devices(1:kidmp_ndevices) (or the environment variable
! KIDMP_DEVICES = "0,1,2,..."): thompson_init builds the tables on every device and mp_thompson_batch spreads the
! columns over them in contiguous ranges (kidmp_batch_step_host_multi: one pipeline per device, the domain sums of the
! surface precipitation reduced with RCCL and left in kidmp_precip_sums).  'p64' arithmetic only.
!
! Side effects kept: the 36 process-rate diagnostics the reference emits from inside mp_thompson
! (M:2962-3124) are replayed after the batched call through KiD's own save_dg, same names, same
! order (per column, per level: 30 mixed-phase rates unless iiwarm, then 6 warm ones), same
! nx == 1 / nx > 1 call forms, and none for a column that took the no_micro early return
! (M:1540).  l_rate_diagnostics = .false. switches them off (big batches: the rate buffer is
! 36 profiles per column).  kidmp_device (or the environment variable KIDMP_DEVICE) selects the GPU.
!
module module_mp_thompson09n

  use iso_c_binding
  use switches, only: l_sediment, l_reuse_thompson_lookup
  use namelists, only: iiwarm, set_Nc
  use parameters, only: nx
  use diagnostics, only: save_dg, i_dgtime

  implicit none
  private

  public :: thompson_init, mp_thompson, mp_thompson_batch, mp_thompson_staging, thompson_finalize
  public :: kidmp_precip_sums_valid
  logical, public :: is_aerosol_aware = .false.          ! M:28 (read at thompson_init)
  logical, public :: l_rate_diagnostics = .true.         ! replay the save_dg calls of M:2962-3124
  integer, public :: kidmp_device = 0                    ! HIP device ordinal of this process (one GPU)
  integer, public :: kidmp_ndevices = 1                  ! > 1: the columns are spread over kidmp_devices(1:kidmp_ndevices)
  integer, public :: kidmp_devices(8) = (/ 0, 1, 2, 3, 4, 5, 6, 7 /)
  ! domain sums (rain, snow, graupel, ice) of ppt after the last mp_thompson_batch call on several devices: the
  ! numerators of the nx-means of W:248-303, reduced over the devices with RCCL inside the library
  real(c_double), public :: kidmp_precip_sums(4) = 0.0_c_double
  character(64), public :: kidmp_cache_dir = 'run_data'  ! where the reference keeps its table cache (M:3710, M:3857)
  character(4), public :: kidmp_arith = 'p64 '           ! 'p64', or with 4-byte default REAL 'p32n' / 'f32'

  ! the diagnosed rates in the reference's emission order (M:2967-3119): 30 mixed-phase, then 6 warm
  integer, parameter :: NRATES = 36, NRATES_MIXED = 30
  character(7), parameter :: rate_names(NRATES) = (/ &
       'pri_inu', 'pri_ide', 'prs_ide', 'prs_sde', 'prg_gde', 'pri_wfz', 'prs_scw', 'prg_scw', 'prg_gcw', 'pri_ihm', &
       'pri_rfz', 'prs_iau', 'prs_sci', 'pri_rci', 'pni_inu', 'pni_ihm', 'pni_wfz', 'pni_rfz', 'pni_ide', 'pni_iau', &
       'pni_sci', 'pni_rci', 'prr_sml', 'prr_gml', 'pnr_rcs', 'pnr_rcg', 'pnr_rci', 'pnr_sml', 'pnr_gml', 'pnr_rfz', &
       'prr_wau', 'prr_rcw', 'prv_rev', 'pnr_wau', 'pnr_rev', 'pnr_rcr' /)

  type, bind(C) :: kidmp_cfg
     integer(c_int32_t) :: iiwarm
     integer(c_int32_t) :: l_sediment
     real(c_double)     :: set_Nc
     integer(c_int32_t) :: device
     integer(c_int32_t) :: is_aerosol_aware
  end type kidmp_cfg

  type(c_ptr), save :: ctx = c_null_ptr                  ! the context (of the first device, when there are several)
  type(c_ptr), save :: mctx = c_null_ptr                 ! kidmp_multi handle, when kidmp_ndevices > 1
  ! Staging arrays of mp_thompson_batch: page-locked (kidmp_host_alloc) and kept between calls, so that the library's
  ! upload / step / download pipeline can move them by DMA.  1 = state (12 profiles), 2 = p, w, dz, 3 = ppt,
  ! 4 = the 36 rate profiles, 5 = the substep counts.
  type(c_ptr), save :: hbuf(5) = c_null_ptr
  integer(c_size_t), save :: hbytes(5) = 0_c_size_t

  interface
     integer(c_int) function kidmp_init(cfg, ctx_out) bind(C, name='kidmp_init')
       import :: c_int, c_ptr, kidmp_cfg
       type(kidmp_cfg), intent(in) :: cfg
       type(c_ptr), intent(out) :: ctx_out
     end function kidmp_init
     subroutine kidmp_finalize(ctx) bind(C, name='kidmp_finalize')
       import :: c_ptr
       type(c_ptr), value :: ctx
     end subroutine kidmp_finalize
     function kidmp_last_error(ctx) result(msg) bind(C, name='kidmp_last_error')
       import :: c_ptr
       type(c_ptr), value :: ctx
       type(c_ptr) :: msg
     end function kidmp_last_error
     ! arrays by address: the ones KiD never fills (nc, nwfa, nifa, w; the frozen species of a warm run) may be NULL
     integer(c_int) function kidmp_batch_step_host_diag(ctx, ncol, nz, dt, qv, qc, qi, qr, qs, qg, ni, nr, &
          nc, nwfa, nifa, t, p, w, dz, ppt, rates, nstep) bind(C, name='kidmp_batch_step_host_diag')
       import :: c_int, c_int32_t, c_int64_t, c_double, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int64_t), value :: ncol
       integer(c_int32_t), value :: nz
       real(c_double), value :: dt
       type(c_ptr), value :: qv, qc, qi, qr, qs, qg, ni, nr, nc, nwfa, nifa, t, p, w, dz, ppt   ! real(c_double) [ncol][nz]; ppt [ncol][4]
       type(c_ptr), value :: rates, nstep            ! NULL, or [ncol][36][nz] doubles / [ncol][4] int32
     end function kidmp_batch_step_host_diag
     integer(c_int) function kidmp_init_multi(cfg, ndev, devices, m_out) bind(C, name='kidmp_init_multi')
       import :: c_int, c_int32_t, c_ptr, kidmp_cfg
       type(kidmp_cfg), intent(in) :: cfg
       integer(c_int32_t), value :: ndev
       integer(c_int32_t), intent(in) :: devices(*)
       type(c_ptr), intent(out) :: m_out
     end function kidmp_init_multi
     subroutine kidmp_finalize_multi(m) bind(C, name='kidmp_finalize_multi')
       import :: c_ptr
       type(c_ptr), value :: m
     end subroutine kidmp_finalize_multi
     type(c_ptr) function kidmp_multi_context(m, i) bind(C, name='kidmp_multi_context')
       import :: c_ptr, c_int32_t
       type(c_ptr), value :: m
       integer(c_int32_t), value :: i
     end function kidmp_multi_context
     function kidmp_multi_last_error(m) result(msg) bind(C, name='kidmp_multi_last_error')
       import :: c_ptr
       type(c_ptr), value :: m
       type(c_ptr) :: msg
     end function kidmp_multi_last_error
     integer(c_int) function kidmp_batch_step_host_multi(m, ncol, nz, dt, qv, qc, qi, qr, qs, qg, ni, nr, &
          nc, nwfa, nifa, t, p, w, dz, ppt, rates, nstep, precip_sums) bind(C, name='kidmp_batch_step_host_multi')
       import :: c_int, c_int32_t, c_int64_t, c_double, c_ptr
       type(c_ptr), value :: m
       integer(c_int64_t), value :: ncol
       integer(c_int32_t), value :: nz
       real(c_double), value :: dt
       type(c_ptr), value :: qv, qc, qi, qr, qs, qg, ni, nr, nc, nwfa, nifa, t, p, w, dz, ppt, rates, nstep
       real(c_double), intent(out) :: precip_sums(4)
     end function kidmp_batch_step_host_multi
     integer(c_int) function kidmp_table_cache_reuse(ctx, dir, l_reuse, write_if_built, status) &
          bind(C, name='kidmp_table_cache_reuse')
       import :: c_int, c_int32_t, c_ptr, c_char
       type(c_ptr), value :: ctx
       character(kind=c_char), intent(in) :: dir(*)
       integer(c_int32_t), value :: l_reuse, write_if_built
       integer(c_int32_t), intent(out) :: status
     end function kidmp_table_cache_reuse
     type(c_ptr) function kidmp_host_alloc(bytes) bind(C, name='kidmp_host_alloc')   ! page-locked host memory
       import :: c_ptr, c_size_t
       integer(c_size_t), value :: bytes
     end function kidmp_host_alloc
     subroutine kidmp_host_free(p) bind(C, name='kidmp_host_free')
       import :: c_ptr
       type(c_ptr), value :: p
     end subroutine kidmp_host_free
     integer(c_int) function kidmp32_batch_step_host(ctx, ncol, nz, dt, qv, qc, qi, qr, qs, qg, ni, nr, &
          nc, nwfa, nifa, t, p, w, dz, ppt, rates, nstep, arith) bind(C, name='kidmp32_batch_step_host')
       import :: c_int, c_int32_t, c_int64_t, c_float, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int64_t), value :: ncol
       integer(c_int32_t), value :: nz, arith
       real(c_float), value :: dt
       type(c_ptr), value :: qv, qc, qi, qr, qs, qg, ni, nr, nc, nwfa, nifa, t, p, w, dz, ppt   ! real(c_float)
       type(c_ptr), value :: rates, nstep            ! NULL, or [ncol][36][nz] doubles / [ncol][4] int32
     end function kidmp32_batch_step_host
  end interface

contains

  ! .true. when mp_thompson_batch ran on several devices and kidmp_precip_sums holds the RCCL-reduced domain sums
  logical function kidmp_precip_sums_valid()
    kidmp_precip_sums_valid = c_associated(mctx)
  end function kidmp_precip_sums_valid

  subroutine stop_on_error(rc, where)
    integer(c_int), intent(in) :: rc
    character(*), intent(in) :: where
    character(kind=c_char), pointer :: cmsg(:)
    type(c_ptr) :: p
    integer :: n
    if (rc == 0) return
    if (c_associated(mctx)) then
       p = kidmp_multi_last_error(mctx)
    else
       p = kidmp_last_error(ctx)
    end if
    write(*,'(a,a,a,i0)') ' module_mp_thompson09n: ', where, ' failed, kidmp code ', rc
    if (c_associated(p)) then
       call c_f_pointer(p, cmsg, [512])
       n = 1
       do while (n < 512 .and. cmsg(n) /= c_null_char)
          n = n + 1
       end do
       write(*,'(1x,512a1)') cmsg(1:n-1)
    end if
    stop 1      ! the reference aborts on init failure too (Fortran runtime error at M:3718)
  end subroutine stop_on_error

  ! hbuf(i) with room for `need` bytes (grown, never shrunk)
  subroutine staging(i, need)
    integer, intent(in) :: i
    integer(c_size_t), intent(in) :: need
    if (need <= hbytes(i) .and. c_associated(hbuf(i))) return
    if (c_associated(hbuf(i))) call kidmp_host_free(hbuf(i))
    hbuf(i) = kidmp_host_alloc(need)
    hbytes(i) = need
    if (.not. c_associated(hbuf(i))) then
       write(*,'(a,i0,a)') ' module_mp_thompson09n: kidmp_host_alloc(', need, ') failed'
       stop 1
    end if
  end subroutine staging

  ! thompson_init, M:374-797: constants on the host, lookup tables built on the GPU (on every GPU of the device list).
  subroutine thompson_init
    type(kidmp_cfg) :: cfg
    integer(c_int) :: rc
    integer(c_int32_t) :: devs(8), status, reuse, wr
    type(c_ptr) :: c
    character(64) :: envdev
    character(kind=c_char) :: cdir(65)
    integer :: envstat, i, n, pos, nxt
    if (c_associated(ctx)) return                         ! micro_init guard, M:384-389
    cfg%iiwarm = merge(1_c_int32_t, 0_c_int32_t, iiwarm)
    cfg%l_sediment = merge(1_c_int32_t, 0_c_int32_t, l_sediment)
    cfg%set_Nc = real(set_Nc, c_double)
    call get_environment_variable('KIDMP_DEVICE', envdev, status=envstat)
    if (envstat == 0 .and. len_trim(envdev) > 0) read(envdev, *, iostat=envstat) kidmp_device
    call get_environment_variable('KIDMP_DEVICES', envdev, status=envstat)    ! "0,1,2,3": several GPUs
    if (envstat == 0 .and. len_trim(envdev) > 0) then
       n = 0;  pos = 1
       do while (pos <= len_trim(envdev) .and. n < 8)
          nxt = index(envdev(pos:), ',')
          if (nxt == 0) nxt = len_trim(envdev) - pos + 2
          n = n + 1
          read(envdev(pos:pos+nxt-2), *, iostat=envstat) kidmp_devices(n)
          if (envstat /= 0) then
             write(*,'(2a)') ' module_mp_thompson09n: cannot read KIDMP_DEVICES=', trim(envdev)
             stop 1
          end if
          pos = pos + nxt
       end do
       kidmp_ndevices = n
       ! a list with ONE entry names the card of the single-device path (KIDMP_DEVICES=3 must not land on GPU 0)
       if (n == 1) kidmp_device = kidmp_devices(1)
    end if
    cfg%device = int(kidmp_device, c_int32_t)
    cfg%is_aerosol_aware = merge(1_c_int32_t, 0_c_int32_t, is_aerosol_aware)
    if (kidmp_ndevices > 1) then
       if (kidmp_ndevices > 8) then
          write(*,'(a)') ' module_mp_thompson09n: at most 8 devices'
          stop 1
       end if
       if (trim(kidmp_arith) /= 'p64') then
          write(*,'(a)') ' module_mp_thompson09n: several devices need kidmp_arith = p64'
          stop 1
       end if
       devs(1:kidmp_ndevices) = int(kidmp_devices(1:kidmp_ndevices), c_int32_t)
       rc = kidmp_init_multi(cfg, int(kidmp_ndevices, c_int32_t), devs, mctx)
       call stop_on_error(rc, 'thompson_init (kidmp_init_multi)')
       ctx = kidmp_multi_context(mctx, 0_c_int32_t)
    else
       rc = kidmp_init(cfg, ctx)
       call stop_on_error(rc, 'thompson_init')
    end if
    ! ---- the reference's table cache, M:3717-3729 / M:3822-3829 and M:3864-3895 / M:4065-4078: per file, read it if
    !      it exists and l_reuse_thompson_lookup is set, else write the freshly built tables (first device only) ----
    if (.not. iiwarm) then
       n = len_trim(kidmp_cache_dir)
       do i = 1, n
          cdir(i) = kidmp_cache_dir(i:i)
       end do
       cdir(n+1) = c_null_char
       reuse = merge(1_c_int32_t, 0_c_int32_t, l_reuse_thompson_lookup)
       do i = 1, max(1, kidmp_ndevices)
          c = ctx
          if (c_associated(mctx)) c = kidmp_multi_context(mctx, int(i-1, c_int32_t))
          wr = merge(1_c_int32_t, 0_c_int32_t, i == 1)
          rc = kidmp_table_cache_reuse(c, cdir, reuse, wr, status)
          call stop_on_error(rc, 'thompson_init (table cache)')
          if (i == 1 .and. iand(status, 2_c_int32_t) /= 0) then          ! the reference's notice: printed by qr_acr_qs only
             ! (M:3872-3881; qr_acr_qg reads its file silently, M:3721-3727), i.e. when the racs file was read (bit 1)
             write(6,*) ' !!!!!!!!!!!!!!!!!! WARNING !!!!!!!!!!!!!!!!!!!'
             write(6,*) ' Reading in pre-calculated lookup tables in    '
             write(6,*) ' Thompson scheme'
             write(6,*) ' If you have changed any microphysical '
             write(6,*) ' parameters, you may need to recalculate these. '
             write(6,*) ' !!!!!!!!!!!!!!!!!! WARNING !!!!!!!!!!!!!!!!!!!'
          end if
       end do
    end if
  end subroutine thompson_init

  subroutine thompson_finalize
    integer :: i
    do i = 1, 5
       if (c_associated(hbuf(i))) call kidmp_host_free(hbuf(i))
       hbuf(i) = c_null_ptr;  hbytes(i) = 0_c_size_t
    end do
    if (c_associated(mctx)) then
       call kidmp_finalize_multi(mctx)                    ! finalises every context, ctx among them
    else if (c_associated(ctx)) then
       call kidmp_finalize(ctx)
    end if
    ctx = c_null_ptr;  mctx = c_null_ptr
  end subroutine thompson_finalize

  ! mp_thompson, M:1156-1177: one column, reference dummy list.
  subroutine mp_thompson (qv1d, qc1d, qi1d, qr1d, qs1d, qg1d, ni1d, &
       nr1d, nc1d, nwfa1d, nifa1d, t1d, p1d, w1d, dzq, &
       pptrain, pptsnow, pptgraul, pptice, &
       kts, kte, dt, ii, jj)
    integer, intent(in) :: kts, kte, ii, jj
    real, dimension(kts:kte), intent(inout) :: qv1d, qc1d, qi1d, qr1d, qs1d, qg1d, ni1d, &
         nr1d, nc1d, nwfa1d, nifa1d, t1d
    real, dimension(kts:kte), intent(in) :: p1d, w1d, dzq
    real, intent(inout) :: pptrain, pptsnow, pptgraul, pptice
    real, intent(in) :: dt
    real :: ppt(4,1)
    integer :: nz
    nz = kte - kts + 1
    ppt(:,1) = (/ pptrain, pptsnow, pptgraul, pptice /)
    call mp_thompson_batch(1, nz, dt, qv1d, qc1d, qi1d, qr1d, qs1d, qg1d, ni1d, nr1d, nc1d, nwfa1d, &
         nifa1d, t1d, p1d, w1d, dzq, ppt)
    pptrain = ppt(1,1); pptsnow = ppt(2,1); pptgraul = ppt(3,1); pptice = ppt(4,1)
    if (.false.) print *, ii, jj          ! ii, jj are debug-only in the reference (M:1269-1274)
  end subroutine mp_thompson

  ! The page-locked staging arrays themselves, for a caller whose default REAL is the storage type of kidmp_arith
  ! (8-byte REAL with 'p64', 4-byte REAL with 'p32n' / 'f32'): st(nz,ncol,12) in the argument order of mp_thompson
  ! (qv qc qi qr qs qg ni nr nc nwfa nifa t), fo(nz,ncol,3) = p, w, dz, pp(4,ncol).  Filled in place and passed to
  ! mp_thompson_batch -- ALL of them, slot by slot -- they are not copied again, in or out.  ok = .false. (and null
  ! pointers) when the kinds differ: the caller then brings its own arrays and mp_thompson_batch converts.
  subroutine mp_thompson_staging(ncol, nz, st, fo, pp, ok)
    integer, intent(in) :: ncol, nz
    real, pointer, intent(out) :: st(:,:,:), fo(:,:,:), pp(:,:)
    logical, intent(out) :: ok
    integer(c_size_t) :: nprof, esize
    nullify(st, fo, pp)
    ok = (kind(1.0) == c_double .and. trim(kidmp_arith) == 'p64') .or. &
         (kind(1.0) == c_float .and. trim(kidmp_arith) /= 'p64')
    if (.not. ok) return
    nprof = int(nz, c_size_t) * int(ncol, c_size_t)
    esize = int(storage_size(1.0) / 8, c_size_t)
    call staging(1, esize * 12 * nprof);  call staging(2, esize * 3 * nprof);  call staging(3, esize * 4 * ncol)
    call c_f_pointer(hbuf(1), st, [nz, ncol, 12]);  call c_f_pointer(hbuf(2), fo, [nz, ncol, 3]);  call c_f_pointer(hbuf(3), pp, [4, ncol])
  end subroutine mp_thompson_staging

  ! ncol columns in one launch.  Arrays are (nz, ncol), k fastest -- KiD's own
  ! storage order -- and ppt is (4, ncol) = rain, snow, graupel, ice, accumulated.
  ! What KiD never fills may be left out (keyword call): nc, nwfa, nifa and w without is_aerosol_aware (W:36 passes
  ! them unset; the library forms the non-aerosol defaults of M:958-964 on the GPU), qi, qs, qg, ni in an iiwarm run
  ! (they stay zero, W:46-52).  Absent arrays are neither staged nor sent across PCIe.
  subroutine mp_thompson_batch(ncol, nz, dt, qv, qc, qi, qr, qs, qg, ni, nr, nc, nwfa, nifa, t, p, w, dz, ppt)
    integer, intent(in) :: ncol, nz
    real, intent(in) :: dt
    real, dimension(nz,ncol), intent(inout), target :: qv, qc, qr, nr, t
    real, dimension(nz,ncol), intent(inout), optional, target :: qi, qs, qg, ni, nc, nwfa, nifa
    real, dimension(nz,ncol), intent(in), target :: p, dz
    real, dimension(nz,ncol), intent(in), optional, target :: w
    real, dimension(4,ncol), intent(inout), target :: ppt
    real(c_double), pointer :: s(:,:,:), f(:,:,:), pp(:,:), rates(:,:,:)
    real(c_float), pointer :: s4(:,:,:), f4(:,:,:), pp4(:,:)
    integer(c_int32_t), pointer :: nstep(:,:)
    type(c_ptr) :: prates, pnstep, ps(12), pf(3)
    integer(c_size_t) :: nprof
    integer(c_int) :: rc
    integer(c_int32_t) :: arith
    logical :: have_frz, have_aer, inplace
    integer :: i, k, r, r0
    if (.not. c_associated(ctx)) call thompson_init
    have_frz = present(qi);  have_aer = present(nc)
    if ((have_frz .neqv. present(qs)) .or. (have_frz .neqv. present(qg)) .or. (have_frz .neqv. present(ni)) .or. &
        (have_aer .neqv. present(nwfa)) .or. (have_aer .neqv. present(nifa))) then
       write(*,'(a)') ' module_mp_thompson09n: qi, qs, qg, ni (and nc, nwfa, nifa) must be passed or left out together'
       stop 1
    end if
    prates = c_null_ptr;  pnstep = c_null_ptr
    nprof = int(nz, c_size_t) * int(ncol, c_size_t)
    if (l_rate_diagnostics) then
       call staging(4, 8_c_size_t * NRATES * nprof)
       call staging(5, 16_c_size_t * ncol)
       call c_f_pointer(hbuf(4), rates, [nz, NRATES, ncol])  ! the C ABI's [ncol][36][nz] / [ncol][4]
       call c_f_pointer(hbuf(5), nstep, [4, ncol])
       prates = hbuf(4);  pnstep = hbuf(5)
    end if
    ps = c_null_ptr;  pf = c_null_ptr
    if (trim(kidmp_arith) /= 'p64') then
       ! ---- binary32 state straight to the GPU: the reference's own REAL / DOUBLE PRECISION split, or all binary32 ----
       if (kind(qv) /= c_float) then
          write(*,'(3a)') ' module_mp_thompson09n: kidmp_arith=', trim(kidmp_arith), ' needs KiD built with 4-byte default REAL'
          stop 1
       end if
       arith = 0_c_int32_t
       if (trim(kidmp_arith) == 'f32') arith = 1_c_int32_t
       call staging(1, 4_c_size_t * 12 * nprof);  call staging(2, 4_c_size_t * 3 * nprof);  call staging(3, 16_c_size_t * ncol)
       call c_f_pointer(hbuf(1), s4, [nz, ncol, 12]);  call c_f_pointer(hbuf(2), f4, [nz, ncol, 3]);  call c_f_pointer(hbuf(3), pp4, [4, ncol])
       ! (an argument that IS its staging slot -- mp_thompson_staging -- needs no copy, in or out)
       inplace = c_associated(c_loc(qv), c_loc(s4(1,1,1))) .and. c_associated(c_loc(qc), c_loc(s4(1,1,2))) .and. &
            c_associated(c_loc(qr), c_loc(s4(1,1,4))) .and. c_associated(c_loc(nr), c_loc(s4(1,1,8))) .and. &
            c_associated(c_loc(t), c_loc(s4(1,1,12))) .and. c_associated(c_loc(p), c_loc(f4(1,1,1))) .and. &
            c_associated(c_loc(dz), c_loc(f4(1,1,3))) .and. c_associated(c_loc(ppt), c_loc(pp4(1,1)))
       if (inplace .and. have_frz) inplace = c_associated(c_loc(qi), c_loc(s4(1,1,3))) .and. c_associated(c_loc(qs), c_loc(s4(1,1,5))) &
            .and. c_associated(c_loc(qg), c_loc(s4(1,1,6))) .and. c_associated(c_loc(ni), c_loc(s4(1,1,7)))
       if (inplace .and. have_aer) inplace = c_associated(c_loc(nc), c_loc(s4(1,1,9))) .and. c_associated(c_loc(nwfa), c_loc(s4(1,1,10))) &
            .and. c_associated(c_loc(nifa), c_loc(s4(1,1,11)))
       if (inplace .and. present(w)) inplace = c_associated(c_loc(w), c_loc(f4(1,1,2)))
       if (.not. inplace) then
       s4(:,:,1) = qv;  s4(:,:,2) = qc;  s4(:,:,4) = qr;  s4(:,:,8) = nr;  s4(:,:,12) = t
       if (have_frz) then
          s4(:,:,3) = qi;  s4(:,:,5) = qs;  s4(:,:,6) = qg;  s4(:,:,7) = ni
       end if
       if (have_aer) then
          s4(:,:,9) = nc;  s4(:,:,10) = nwfa;  s4(:,:,11) = nifa
       end if
       f4(:,:,1) = p;  f4(:,:,3) = dz
       if (present(w)) f4(:,:,2) = w
       pp4 = ppt
       end if
       do i = 1, 12
          ps(i) = c_loc(s4(1,1,i))
       end do
       pf(1) = c_loc(f4(1,1,1));  pf(3) = c_loc(f4(1,1,3))
       if (present(w)) pf(2) = c_loc(f4(1,1,2))
       if (.not. have_frz) then
          ps(3) = c_null_ptr;  ps(5) = c_null_ptr;  ps(6) = c_null_ptr;  ps(7) = c_null_ptr
       end if
       if (.not. have_aer) ps(9:11) = c_null_ptr
       rc = kidmp32_batch_step_host(ctx, int(ncol, c_int64_t), int(nz, c_int32_t), real(dt, c_float), &
            ps(1), ps(2), ps(3), ps(4), ps(5), ps(6), ps(7), ps(8), ps(9), ps(10), ps(11), ps(12), &
            pf(1), pf(2), pf(3), c_loc(pp4), prates, pnstep, arith)
       call stop_on_error(rc, 'mp_thompson')
       if (.not. inplace) then
       qv = s4(:,:,1);  qc = s4(:,:,2);  qr = s4(:,:,4);  nr = s4(:,:,8);  t = s4(:,:,12)
       if (have_frz) then
          qi = s4(:,:,3);  qs = s4(:,:,5);  qg = s4(:,:,6);  ni = s4(:,:,7)
       end if
       if (have_aer) then
          nc = s4(:,:,9);  nwfa = s4(:,:,10);  nifa = s4(:,:,11)
       end if
       ppt = pp4
       end if
    else
    call staging(1, 8_c_size_t * 12 * nprof);  call staging(2, 8_c_size_t * 3 * nprof);  call staging(3, 32_c_size_t * ncol)
    call c_f_pointer(hbuf(1), s, [nz, ncol, 12]);  call c_f_pointer(hbuf(2), f, [nz, ncol, 3]);  call c_f_pointer(hbuf(3), pp, [4, ncol])
    inplace = c_associated(c_loc(qv), c_loc(s(1,1,1))) .and. c_associated(c_loc(qc), c_loc(s(1,1,2))) .and. &
         c_associated(c_loc(qr), c_loc(s(1,1,4))) .and. c_associated(c_loc(nr), c_loc(s(1,1,8))) .and. &
         c_associated(c_loc(t), c_loc(s(1,1,12))) .and. c_associated(c_loc(p), c_loc(f(1,1,1))) .and. &
         c_associated(c_loc(dz), c_loc(f(1,1,3))) .and. c_associated(c_loc(ppt), c_loc(pp(1,1)))
    if (inplace .and. have_frz) inplace = c_associated(c_loc(qi), c_loc(s(1,1,3))) .and. c_associated(c_loc(qs), c_loc(s(1,1,5))) &
         .and. c_associated(c_loc(qg), c_loc(s(1,1,6))) .and. c_associated(c_loc(ni), c_loc(s(1,1,7)))
    if (inplace .and. have_aer) inplace = c_associated(c_loc(nc), c_loc(s(1,1,9))) .and. c_associated(c_loc(nwfa), c_loc(s(1,1,10))) &
         .and. c_associated(c_loc(nifa), c_loc(s(1,1,11)))
    if (inplace .and. present(w)) inplace = c_associated(c_loc(w), c_loc(f(1,1,2)))
    if (.not. inplace) then
    s(:,:,1) = qv;  s(:,:,2) = qc;  s(:,:,4) = qr;  s(:,:,8) = nr;  s(:,:,12) = t
    if (have_frz) then
       s(:,:,3) = qi;  s(:,:,5) = qs;  s(:,:,6) = qg;  s(:,:,7) = ni
    end if
    if (have_aer) then
       s(:,:,9) = nc;  s(:,:,10) = nwfa;  s(:,:,11) = nifa
    end if
    f(:,:,1) = p;  f(:,:,3) = dz
    if (present(w)) f(:,:,2) = w
    pp = ppt
    end if
    do i = 1, 12
       ps(i) = c_loc(s(1,1,i))
    end do
    pf(1) = c_loc(f(1,1,1));  pf(3) = c_loc(f(1,1,3))
    if (present(w)) pf(2) = c_loc(f(1,1,2))
    if (.not. have_frz) then
       ps(3) = c_null_ptr;  ps(5) = c_null_ptr;  ps(6) = c_null_ptr;  ps(7) = c_null_ptr
    end if
    if (.not. have_aer) ps(9:11) = c_null_ptr
    if (c_associated(mctx)) then                           ! several GPUs: contiguous column ranges, one pipeline each
       rc = kidmp_batch_step_host_multi(mctx, int(ncol, c_int64_t), int(nz, c_int32_t), real(dt, c_double), &
            ps(1), ps(2), ps(3), ps(4), ps(5), ps(6), ps(7), ps(8), ps(9), ps(10), ps(11), ps(12), &
            pf(1), pf(2), pf(3), c_loc(pp), prates, pnstep, kidmp_precip_sums)
    else
       rc = kidmp_batch_step_host_diag(ctx, int(ncol, c_int64_t), int(nz, c_int32_t), real(dt, c_double), &
            ps(1), ps(2), ps(3), ps(4), ps(5), ps(6), ps(7), ps(8), ps(9), ps(10), ps(11), ps(12), &
            pf(1), pf(2), pf(3), c_loc(pp), prates, pnstep)
    end if
    call stop_on_error(rc, 'mp_thompson')
    if (.not. inplace) then
    qv = s(:,:,1);  qc = s(:,:,2);  qr = s(:,:,4);  nr = s(:,:,8);  t = s(:,:,12)
    if (have_frz) then
       qi = s(:,:,3);  qs = s(:,:,5);  qg = s(:,:,6);  ni = s(:,:,7)
    end if
    if (have_aer) then
       nc = s(:,:,9);  nwfa = s(:,:,10);  nifa = s(:,:,11)
    end if
    ppt = pp
    end if
    end if
    ! ---- the KiD block of M:2962-3124: per column, per level, 30 mixed-phase rates (.not. iiwarm) then 6 warm
    !      ones; save_dg(k, value, ...) when nx == 1, save_dg(k, ii, value, ...) otherwise; a column that left
    !      through the no_micro return (M:1540: all four substep counts 0) never reached the block ----
    if (l_rate_diagnostics) then
       r0 = 1
       if (iiwarm) r0 = NRATES_MIXED + 1
       do i = 1, ncol
          if (all(nstep(:,i) == 0)) cycle
          do k = 1, nz
             do r = r0, NRATES
                if (nx == 1) then
                   call save_dg(k, rates(k,r,i), rate_names(r), i_dgtime, units='/kg/s', dim='z')
                else
                   call save_dg(k, i, rates(k,r,i), rate_names(r), i_dgtime, units='/kg/s', dim='z')
                end if
             end do
          end do
       end do
    end if
  end subroutine mp_thompson_batch

end module module_mp_thompson09n
