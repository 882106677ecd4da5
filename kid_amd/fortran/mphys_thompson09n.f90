!
! mphys_thompson09n -- drop-in replacement for the KiD adapter of the same name
! (/root/reference/mphys_thompson09n.f90, "W:").  Same module name, same public
! entry `mphys_thompson09_interfacen` (no arguments; all data through KiD's
! column_variables / parameters / physconst modules, W:11-17), same outputs:
! d*_mphys tendencies (W:198-245) and the surface-precipitation diagnostics
! through save_dg (W:155-192, W:248-303).
!
! What changes: the `do i=1,nx` loop around mp_thompson (W:54-246) becomes one
! batched call -- all nx columns go to the MI355X in a single launch -- and the
! inputs the reference leaves unset (nc1d, nwfa1d, nifa1d, w1d; W:36) get the
! scheme's own non-aerosol defaults (M:958-964).
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

  logical :: micro_unset=.True.
  integer:: ih, imom
  character(max_char_len) :: name, units

contains

  Subroutine mphys_thompson09_interfacen

    real :: t2d(nz,nx), p2d(nz,nx), dz2d(nz,nx), w2d(nz,nx), qv2d(nz,nx), qc2d(nz,nx), qr2d(nz,nx), &
         nr2d(nz,nx), qi2d(nz,nx), ni2d(nz,nx), qs2d(nz,nx), qg2d(nz,nx), nc2d(nz,nx), nwfa2d(nz,nx), &
         nifa2d(nz,nx)
    real :: ppt(4,nx), rho
    real :: pptrain_2d(nx), pptsnow_2d(nx), pptgraul_2d(nx), pptice_2d(nx)
    integer :: i, k

    qi2d = 0.0; ni2d = 0.0; qs2d = 0.0; qg2d = 0.0          ! W:46-52
    w2d = 0.0
    ppt = 0.0                                                 ! W:55-58

    do i=1,nx                                                 ! gather, W:59-97
       do k=1,nz
          t2d(k,i) = (theta(k,i) + (dtheta_adv(k,i)+dtheta_div(k,i))*dt )*exner(k,i)
          p2d(k,i) = p0*exner(k,i)**(1./r_on_cp)
          dz2d(k,i) = dz(k)
          qv2d(k,i) = qv(k,i) + (dqv_adv(k,i)+dqv_div(k,i))*dt
          qc2d(k,i) = hydrometeors(k,i,1)%moments(1,1) &
               + (dhydrometeors_adv(k,i,1)%moments(1,1) + dhydrometeors_div(k,i,1)%moments(1,1))*dt
          qr2d(k,i) = hydrometeors(k,i,2)%moments(1,1) &
               + (dhydrometeors_adv(k,i,2)%moments(1,1) + dhydrometeors_div(k,i,2)%moments(1,1))*dt
          nr2d(k,i) = hydrometeors(k,i,2)%moments(1,2) &
               + (dhydrometeors_adv(k,i,2)%moments(1,2) + dhydrometeors_div(k,i,2)%moments(1,2))*dt
          if (.not. iiwarm) then
             qi2d(k,i) = hydrometeors(k,i,3)%moments(1,1) &
                  + (dhydrometeors_adv(k,i,3)%moments(1,1) + dhydrometeors_div(k,i,3)%moments(1,1))*dt
             ni2d(k,i) = hydrometeors(k,i,3)%moments(1,2) &
                  + (dhydrometeors_adv(k,i,3)%moments(1,2) + dhydrometeors_div(k,i,3)%moments(1,2))*dt
             qs2d(k,i) = hydrometeors(k,i,4)%moments(1,1) &
                  + (dhydrometeors_adv(k,i,4)%moments(1,1) + dhydrometeors_div(k,i,4)%moments(1,1))*dt
             qg2d(k,i) = hydrometeors(k,i,5)%moments(1,1) &
                  + (dhydrometeors_adv(k,i,5)%moments(1,1) + dhydrometeors_div(k,i,5)%moments(1,1))*dt
          end if
          ! non-aerosol defaults for what the reference leaves unset (W:36; M:958-964)
          rho = 0.622*p2d(k,i)/(287.04*t2d(k,i)*(qv2d(k,i)+0.622))
          nc2d(k,i) = set_Nc*1.e6/rho
          nwfa2d(k,i) = 11.1E6/rho
          nifa2d(k,i) = 0.5E6*0.01/rho
       end do
    end do

    if (micro_unset) then                                     ! W:100-103
       call thompson_init
       micro_unset=.False.
    end if

    call mp_thompson_batch(nx, nz, dt, qv2d, qc2d, qi2d, qr2d, qs2d, qg2d, ni2d, nr2d, &
         nc2d, nwfa2d, nifa2d, t2d, p2d, w2d, dz2d, ppt)       ! W:143-152, all columns at once

    pptrain_2d = ppt(1,:); pptsnow_2d = ppt(2,:); pptgraul_2d = ppt(3,:); pptice_2d = ppt(4,:)

    if (nx == 1) then                                         ! W:155-182
       imom=1
       ih=2
       name='surface_ppt_for_'//trim(h_names(ih))
       units=trim(mom_units(imom))//' m'
       call save_dg(pptrain_2d(1), name, i_dgtime,  units, dim='time')
       ih=3
       name='surface_ppt_for_'//trim(h_names(ih))
       call save_dg(pptice_2d(1), name, i_dgtime,  units, dim='time')
       ih=4
       name='surface_ppt_for_'//trim(h_names(ih))
       call save_dg(pptsnow_2d(1), name, i_dgtime,  units, dim='time')
       ih=5
       name='surface_ppt_for_'//trim(h_names(ih))
       call save_dg(pptgraul_2d(1), name, i_dgtime,  units, dim='time')
       name='total_surface_ppt'
       call save_dg((pptice_2d(1)+pptrain_2d(1)+pptsnow_2d(1)+pptgraul_2d(1))/nx, name, i_dgtime, &
            units, dim='time')
    end if

    do i=1,nx                                                 ! back out tendencies, W:198-245
       do k=1,nz
          dtheta_mphys(k,i)=(t2d(k,i)/exner(k,i)-theta(k,i))/dt - ( dtheta_adv(k,i)+dtheta_div(k,i))
          dqv_mphys(k,i)=(qv2d(k,i) - qv(k,i))/dt - ( dqv_adv(k,i)+dqv_div(k,i))
          dhydrometeors_mphys(k,i,1)%moments(1,1)= (qc2d(k,i)-hydrometeors(k,i,1)%moments(1,1))/dt &
               - (dhydrometeors_adv(k,i,1)%moments(1,1) + dhydrometeors_div(k,i,1)%moments(1,1))
          dhydrometeors_mphys(k,i,2)%moments(1,1)= (qr2d(k,i)-hydrometeors(k,i,2)%moments(1,1))/dt &
               - (dhydrometeors_adv(k,i,2)%moments(1,1) + dhydrometeors_div(k,i,2)%moments(1,1))
          dhydrometeors_mphys(k,i,2)%moments(1,2)= (nr2d(k,i)-hydrometeors(k,i,2)%moments(1,2))/dt &
               - (dhydrometeors_adv(k,i,2)%moments(1,2) + dhydrometeors_div(k,i,2)%moments(1,2))
          if (.not.iiwarm)then
             dhydrometeors_mphys(k,i,3)%moments(1,1)= (qi2d(k,i)-hydrometeors(k,i,3)%moments(1,1))/dt &
                  - (dhydrometeors_adv(k,i,3)%moments(1,1) + dhydrometeors_div(k,i,3)%moments(1,1))
             dhydrometeors_mphys(k,i,3)%moments(1,2)= (ni2d(k,i)-hydrometeors(k,i,3)%moments(1,2))/dt &
                  - (dhydrometeors_adv(k,i,3)%moments(1,2) + dhydrometeors_div(k,i,3)%moments(1,2))
             dhydrometeors_mphys(k,i,4)%moments(1,1)= (qs2d(k,i)-hydrometeors(k,i,4)%moments(1,1))/dt &
                  - (dhydrometeors_adv(k,i,4)%moments(1,1) + dhydrometeors_div(k,i,4)%moments(1,1))
             dhydrometeors_mphys(k,i,5)%moments(1,1)= (qg2d(k,i)-hydrometeors(k,i,5)%moments(1,1))/dt &
                  - (dhydrometeors_adv(k,i,5)%moments(1,1) + dhydrometeors_div(k,i,5)%moments(1,1))
          end if
       end do
    end do

    if (nx > 1) then                                          ! W:248-303, same call order
       imom=1
       units=trim(mom_units(imom))//' m'
       ! domain means
       ih=2
       name='surface_ppt_for_'//trim(h_names(ih))
       call save_dg(pptrain_2d/nx, name, i_dgtime,  units, dim='time')
       ih=3
       name='surface_ppt_for_'//trim(h_names(ih))
       call save_dg(pptice_2d/nx, name, i_dgtime,  units, dim='time')
       ih=4
       name='surface_ppt_for_'//trim(h_names(ih))
       call save_dg(pptsnow_2d/nx, name, i_dgtime,  units, dim='time')
       ih=5
       name='surface_ppt_for_'//trim(h_names(ih))
       call save_dg(pptgraul_2d/nx, name, i_dgtime,  units, dim='time')
       name='total_surface_ppt'
       call save_dg((pptice_2d+pptrain_2d+pptsnow_2d+pptgraul_2d)/nx, name, i_dgtime, units, dim='time')
       ! all horizontal columns
       ih=2
       name='surface_ppt_for_'//trim(h_names(ih))
       call save_dg(pptrain_2d, name, i_dgtime,  units, dim='time')
       ih=3
       name='surface_ppt_for_'//trim(h_names(ih))
       call save_dg(pptice_2d, name, i_dgtime,  units, dim='time')
       ih=4
       name='surface_ppt_for_'//trim(h_names(ih))
       call save_dg(pptsnow_2d, name, i_dgtime,  units, dim='time')
       ih=5
       name='surface_ppt_for_'//trim(h_names(ih))
       call save_dg(pptgraul_2d, name, i_dgtime,  units, dim='time')
       name='total_surface_ppt'
       call save_dg((pptice_2d+pptrain_2d+pptsnow_2d+pptgraul_2d), name, i_dgtime, units, dim='time')
       ! the reference also saves 'total_ppt_level' from pptrain_2d_prof (W:305-307), an array it
       ! never assigns (W:191 is commented out): not emitted.
    endif

  end Subroutine mphys_thompson09_interfacen

end module mphys_thompson09n
