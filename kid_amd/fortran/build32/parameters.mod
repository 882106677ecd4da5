﻿!mod$ v1 sum:732c0bf8568496f5
module parameters
integer(4),parameter::max_char_len=200_4
integer(4),parameter::nspecies=5_4
integer(4),parameter::num_h_moments(1_8:5_8)=[INTEGER(4)::1_4,2_4,2_4,1_4,1_4]
integer(4),parameter::num_h_bins(1_8:5_8)=[INTEGER(4)::1_4,1_4,1_4,1_4,1_4]
integer(4)::nz
integer(4)::nx
real(4)::dt
character(10_4,1)::h_names(1_8:5_8)
character(10_4,1)::mom_units(1_8:2_8)
end
