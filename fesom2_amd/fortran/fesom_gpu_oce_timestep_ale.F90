! The drop-in under the reference's OWN name.  A FESOM2 build that
!   (1) compiles src/oce_ale.F90 with  -Doce_timestep_ale=oce_timestep_ale_cpu  (one preprocessor definition, no source edit: the
!       reference's routine keeps existing under the new name; nothing in this layer calls it -- a run whose options the library refuses stops with the
!       library's message through status_check, it is never stepped on the CPU silently), and
!   (2) adds fesom_gpu_shim.F90 and this file,
! leaves src/fvom_main.F90 untouched: its  call oce_timestep_ale(n, mesh)  (fvom_main.F90:250) resolves to this subroutine, which
! steps on the MI355X.  The  call compute_vel_nodes(mesh)  of fvom_main.F90:216 may stay: the GPU step forms the nodal velocities itself.
! Exercised by oracle/ref/build_ref.sh (fesom_gpu_dropin.x) + tests/test_gpu_dropin.py.
subroutine oce_timestep_ale(n, mesh)
  use MOD_MESH
  use fesom_gpu_shim
  implicit none
  integer, intent(in) :: n
  type(t_mesh), intent(in), target :: mesh
  call oce_timestep_ale_gpu(n, mesh)
end subroutine oce_timestep_ale
