﻿!mod$ v1 sum:465a5bc2646b2da6
module namelists
logical(4)::iiwarm
real(4)::set_nc
end
