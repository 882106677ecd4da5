﻿!mod$ v1 sum:95f9d21f04c7c2f9
module column_variables
type::species
real(4)::moments(1_8:1_8,1_8:2_8)
end type
real(4),allocatable::theta(:,:)
real(4),allocatable::dtheta_adv(:,:)
real(4),allocatable::dtheta_div(:,:)
real(4),allocatable::dtheta_mphys(:,:)
real(4),allocatable::exner(:,:)
real(4),allocatable::qv(:,:)
real(4),allocatable::dqv_adv(:,:)
real(4),allocatable::dqv_div(:,:)
real(4),allocatable::dqv_mphys(:,:)
real(4),allocatable::dz(:)
type(species),allocatable::hydrometeors(:,:,:)
type(species),allocatable::dhydrometeors_adv(:,:,:)
type(species),allocatable::dhydrometeors_div(:,:,:)
type(species),allocatable::dhydrometeors_mphys(:,:,:)
contains
subroutine alloc_columns(nz,nx)
integer(4),intent(in)::nz
integer(4),intent(in)::nx
end
end
