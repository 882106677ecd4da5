﻿!mod$ v1 sum:ec76163b2e653efc
module typekind
integer(4),parameter::wp=4_4
intrinsic::kind
end
