﻿!mod$ v1 sum:a7514de25d14fde1
module diagnostics
integer(4)::i_dgtime
real(4)::last_scalar(1_8:8_8)
integer(4)::n_scalar
interface save_dg
procedure::save_dg_scalar
procedure::save_dg_1d
end interface
contains
subroutine save_dg_scalar(v,name,it,units,dim)
real(4),intent(in)::v
character(*,1),intent(in)::name
integer(4),intent(in)::it
character(*,1),intent(in)::units
character(*,1),intent(in)::dim
end
subroutine save_dg_1d(v,name,it,units,dim)
real(4),intent(in)::v(:)
character(*,1),intent(in)::name
integer(4),intent(in)::it
character(*,1),intent(in)::units
character(*,1),intent(in)::dim
end
end
