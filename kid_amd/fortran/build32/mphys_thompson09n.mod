﻿!mod$ v1 sum:5a47d1a420dbd21d
!need$ 732c0bf8568496f5 n parameters
!need$ 95f9d21f04c7c2f9 n column_variables
!need$ 63943bc4c828cc73 n physconst
!need$ 31d425976ca3ddc6 n module_mp_thompson09n
!need$ a7514de25d14fde1 n diagnostics
!need$ 465a5bc2646b2da6 n namelists
module mphys_thompson09n
use parameters,only:num_h_moments
use parameters,only:num_h_bins
use parameters,only:nspecies
use parameters,only:nz
use parameters,only:dt
use parameters,only:h_names
use parameters,only:mom_units
use parameters,only:max_char_len
use parameters,only:nx
use physconst,only:p0
use physconst,only:r_on_cp
use physconst,only:pi
use namelists,only:iiwarm
use namelists,only:set_nc
use diagnostics,only:save_dg
use diagnostics,only:i_dgtime
use column_variables,only:species
use column_variables,only:theta
use column_variables,only:dtheta_adv
use column_variables,only:dtheta_div
use column_variables,only:dtheta_mphys
use column_variables,only:exner
use column_variables,only:qv
use column_variables,only:dqv_adv
use column_variables,only:dqv_div
use column_variables,only:dqv_mphys
use column_variables,only:dz
use column_variables,only:hydrometeors
use column_variables,only:dhydrometeors_adv
use column_variables,only:dhydrometeors_div
use column_variables,only:dhydrometeors_mphys
use column_variables,only:alloc_columns
use module_mp_thompson09n,only:is_aerosol_aware
use module_mp_thompson09n,only:thompson_init
use module_mp_thompson09n,only:thompson_finalize
use module_mp_thompson09n,only:mp_thompson
use module_mp_thompson09n,only:mp_thompson_batch
use diagnostics,only:diagnostics$diagnostics$save_dg_1d=>save_dg_1d
use diagnostics,only:diagnostics$diagnostics$save_dg_scalar=>save_dg_scalar
logical(4)::micro_unset
integer(4)::ih
integer(4)::imom
character(200_4,1)::name
character(200_4,1)::units
integer(4),parameter,private::s_qv=1_4
integer(4),parameter,private::s_qc=2_4
integer(4),parameter,private::s_qi=3_4
integer(4),parameter,private::s_qr=4_4
integer(4),parameter,private::s_qs=5_4
integer(4),parameter,private::s_qg=6_4
integer(4),parameter,private::s_ni=7_4
integer(4),parameter,private::s_nr=8_4
integer(4),parameter,private::s_nc=9_4
integer(4),parameter,private::s_nwfa=10_4
integer(4),parameter,private::s_nifa=11_4
integer(4),parameter,private::s_t=12_4
integer(4),parameter,private::nslot=12_4
integer(4),parameter,private::nhyd=7_4
integer(4),parameter,private::hyd_slot(1_8:7_8)=[INTEGER(4)::2_4,4_4,8_4,3_4,7_4,5_4,6_4]
integer(4),parameter,private::hyd_spec(1_8:7_8)=[INTEGER(4)::1_4,2_4,2_4,3_4,3_4,4_4,5_4]
integer(4),parameter,private::hyd_mom(1_8:7_8)=[INTEGER(4)::1_4,1_4,2_4,1_4,2_4,1_4,1_4]
integer(4),parameter,private::dg_spec(1_8:4_8)=[INTEGER(4)::2_4,3_4,4_4,5_4]
integer(4),parameter,private::dg_row(1_8:4_8)=[INTEGER(4)::1_4,4_4,2_4,3_4]
contains
subroutine mphys_thompson09_interfacen()
end
end
