﻿!mod$ v1 sum:72b9b114bcae4a23
module switches
logical(4)::l_sediment
logical(4)::l_reuse_thompson_lookup
end
