﻿!mod$ v1 sum:63943bc4c828cc73
module physconst
real(4)::p0
real(4)::r_on_cp
real(4)::pi
end
