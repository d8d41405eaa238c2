! Host side of the MI355X ocean core in the reference's own language: the thin Fortran layer a FESOM2 build adds to call
! libfesom_gpu.so (include/fesom_gpu.h) in place of its CPU time step.
!
!   call fesom_gpu_setup(mesh)            once, after mesh_setup + ocean_setup (src/fvom_main.F90:92-97)
!   call oce_timestep_ale_gpu(n, mesh)    instead of compute_vel_nodes + oce_timestep_ale(n, mesh)  (fvom_main.F90:216,250)
!   call fesom_gpu_fetch_state(mesh)      when the host needs the fields (output, restart, diagnostics)
!   call fesom_gpu_shutdown()
!
! Everything is handed over by address (ISO_C_BINDING): the mesh from t_mesh (src/MOD_MESH.F90:19-95) and o_MESH/o_ARRAYS
! (src/oce_modules.F90:196-353), the partition from g_PARSUP (src/gen_modules_partitioning.F90), the options from the
! namelist variables of o_PARAM / g_config.  Fortran keeps owning every host array.  Errors follow the reference's
! convention: message on the rank, pe_status = 1, status_check aborts (src/gen_comm.F90:644-657).
! One MPI rank <-> one GPU.  This file compiles against the reference's module files (it `use`s them), nothing else.
module fesom_gpu_shim
  use iso_c_binding
  use o_PARAM
  use MOD_MESH
  use o_MESH
  use o_ARRAYS
  use g_PARSUP
  use g_config
  use g_forcing_arrays, only: real_salt_flux, sw_3d
  use i_ARRAYS, only: u_ice, v_ice, a_ice, m_ice, m_snow, S_oc_array
  use g_forcing_arrays, only: press_air, thdgr
  use i_therm_param, only: Sice
  use g_sbf, only: l_mslp
  implicit none
  private
  public :: fesom_gpu_setup, oce_timestep_ale_gpu, fesom_gpu_fetch_state, fesom_gpu_push_state, fesom_gpu_shutdown, fesom_gpu_profile
  ! .true.: every step runs phase by phase (fesom_gpu_profile_step) and its device times are added to the reference's own phase
  ! statistics rtime_oce* (src/oce_ale.F90:2771-2777, printed by fvom_main's BENCHMARK RUNTIME block); slower than the normal step
  logical, save :: fesom_gpu_profile = .false.

  ! ---- struct layouts = include/fesom_gpu.h, field by field
  type, bind(C) :: fesom_mesh_desc
     integer(c_int) :: nod2D, elem2D, edge2D, edge2D_in, nl
     integer(c_int) :: myDim_nod2D, eDim_nod2D, myDim_elem2D, eDim_elem2D, eXDim_elem2D, myDim_edge2D, eDim_edge2D
     integer(c_int) :: max_nod_in_elem, ssh_nza
     type(c_ptr) :: myList_nod2D, myList_elem2D, myList_edge2D, coord_nod2D, geo_coord_nod2D
     type(c_ptr) :: elem2D_nodes, edges, edge_tri, elem_edges, elem_neighbors, nod_in_elem2D, nod_in_elem2D_num
     type(c_ptr) :: nlevels, ulevels, nlevels_nod2D, ulevels_nod2D, nlevels_nod2D_min, ulevels_nod2D_max
     type(c_ptr) :: zbar, Z, depth, elem_area, area, area_inv, areasvol, areasvol_inv, mesh_resolution
     type(c_ptr) :: gradient_sca, gradient_vec, edge_dxdy, edge_cross_dxdy, elem_cos, metric_factor, coriolis, coriolis_node
     type(c_ptr) :: ssh_rowptr, ssh_colind, ssh_colind_loc, ssh_values, edge_up_dn_tri
     type(c_ptr) :: zbar_n_bot, zbar_n_srf, bottom_node_thickness, zbar_e_bot, zbar_e_srf, bottom_elem_thickness
  end type
  type, bind(C) :: fesom_com_desc
     integer(c_int) :: rPEnum, sPEnum
     type(c_ptr) :: rPE, rptr, rlist, sPE, sptr, slist
  end type
  type, bind(C) :: fesom_part_desc
     integer(c_int) :: npes, mype
     type(fesom_com_desc) :: com_nod2D, com_elem2D, com_elem2D_full
  end type
  type, bind(C) :: fesom_params
     real(c_double) :: dt
     integer(c_int) :: which_ale, use_partial_cell, state_equation, num_tracers, mom_adv, visc_option, i_vert_visc, i_vert_diff, &
                       w_split, mix_scheme, use_instabmix, use_windmix, windmix_nl, toy_soufflet
     real(c_double) :: alpha, theta, epsilon, C_d, A_ver, K_ver, K_hor, gamma0, gamma1, gamma2, easy_bs_return, w_max_cfl, &
                       tra_adv_ph, tra_adv_pv, instabmix_kv, windmix_kv, cyclic_length
     integer(c_int) :: with_diffusion, solver_x0_order, Fer_GM
     real(c_double) :: K_GM_max, K_GM_min
     integer(c_int) :: K_GM_bvref
     real(c_double) :: K_GM_rampmax, K_GM_rampmin, K_GM_resscalorder
     integer(c_int) :: scaling_Ferreira, scaling_Rossby, scaling_resolution, scaling_FESOM14, Redi
     real(c_double) :: visc_sh_limit, diff_sh_limit, Ricr, concv
     integer(c_int) :: use_sw_pene, tra_adv_ver, tra_adv_hor, Kv0_const, solver_precond, tra_adv_lim, solver_xinv_its
     real(c_double) :: Leith_c, Div_c
     integer(c_int) :: which_pgf, use_momix
     real(c_double) :: momix_lat, momix_kv
     integer(c_int) :: use_kpp_nonlclflx, ref_sss_local
     real(c_double) :: ref_sss
     integer(c_int) :: smooth_bh_tra, double_diffusion
     integer(c_int) :: use_floatice, l_mslp, use_global_tides
     real(c_double) :: max_ice_loading
     integer(c_int) :: SPP
     real(c_double) :: Sice, clim_relax
     integer(c_int) :: lzstar_lev
     real(c_double) :: min_hnode
     real(c_double) :: c_back, K_back, uke_scaling_factor, rosb_dis, scale_area
     integer(c_int) :: uke_scaling, smooth_back, smooth_dis, smooth_back_tend
     integer(c_int) :: use_cavity, use_density_ref
     real(c_double) :: density_ref_T, density_ref_S
     integer(c_int) :: use_cavity_partial_cell
  end type
  type, bind(C) :: fesom_state_desc
     type(c_ptr) :: tr_arr, tr_arr_old, UV, UV_rhsAB, eta_n, d_eta, ssh_rhs, ssh_rhs_old, hbar, hbar_old, dhe, hnode, hnode_new, &
                    helem, zbar_3d_n, Z_3d_n, Wvel, Wvel_e, Wvel_i, ssh_values
  end type
  type, bind(C) :: fesom_transport
     type(c_ptr) :: ctx
     type(c_funptr) :: exchange, allreduce_sum
  end type
  type, bind(C) :: fesom_forcing_desc
     type(c_ptr) :: stress_surf, heat_flux, water_flux, virtual_salt, relax_salt, real_salt_flux, stress_atmoce_x, stress_atmoce_y, sw_3d, m_ice, m_snow, press_air, ssh_gp, thdgr, S_oc_array, u_ice, v_ice, a_ice
  end type

  interface
     integer(c_int) function c_fesom_gpu_init(mesh, part, par) bind(C, name='fesom_gpu_init')
       import
       type(fesom_mesh_desc), intent(in) :: mesh
       type(c_ptr), value :: part
       type(fesom_params), intent(in) :: par
     end function
     integer(c_int) function c_fesom_gpu_upload_state(st) bind(C, name='fesom_gpu_upload_state')
       import
       type(fesom_state_desc), intent(in) :: st
     end function
     integer(c_int) function c_fesom_gpu_download_state(st) bind(C, name='fesom_gpu_download_state')
       import
       type(fesom_state_desc), intent(in) :: st
     end function
     integer(c_int) function c_fesom_gpu_get_field(name, out, count) bind(C, name='fesom_gpu_get_field')
       import
       character(kind=c_char), intent(in) :: name(*)
       type(c_ptr), value :: out
       integer(c_long_long), value :: count
     end function
     integer(c_int) function c_fesom_gpu_set_field(name, src, count) bind(C, name='fesom_gpu_set_field')
       import
       character(kind=c_char), intent(in) :: name(*)
       type(c_ptr), value :: src
       integer(c_long_long), value :: count
     end function
     integer(c_int) function c_fesom_gpu_set_forcing(f) bind(C, name='fesom_gpu_set_forcing')
       import
       type(fesom_forcing_desc), intent(in) :: f
     end function
     integer(c_int) function c_fesom_gpu_step(n) bind(C, name='fesom_gpu_step')
       import
       integer(c_int), value :: n
     end function
     integer(c_int) function c_fesom_gpu_step_partitioned(n, t) bind(C, name='fesom_gpu_step_partitioned')
       import
       integer(c_int), value :: n
       type(fesom_transport), intent(in) :: t
     end function
     integer(c_int) function c_fesom_gpu_step_partitioned_builtin(n, t) bind(C, name='fesom_gpu_step_partitioned')
       import
       integer(c_int), value :: n
       type(c_ptr), value :: t                   ! c_null_ptr: the library's built-in RCCL transport
     end function
     integer(c_int) function c_fesom_gpu_comm_unique_id(id128) bind(C, name='fesom_gpu_comm_unique_id')
       import
       character(kind=c_char), intent(out) :: id128(128)
     end function
     integer(c_int) function c_fesom_gpu_comm_init(id128, nranks, rank) bind(C, name='fesom_gpu_comm_init')
       import
       character(kind=c_char), intent(in) :: id128(128)
       integer(c_int), value :: nranks, rank
     end function
     integer(c_int) function c_fesom_gpu_comm_selftest(n) bind(C, name='fesom_gpu_comm_selftest')
       import
       integer(c_int), value :: n
     end function
     integer(c_int) function c_fesom_gpu_copy(dst, src, bytes, dir) bind(C, name='fesom_gpu_copy')
       import
       type(c_ptr), value :: dst, src
       integer(c_long_long), value :: bytes
       integer(c_int), value :: dir              ! 0: device -> host, 1: host -> device
     end function
     integer(c_int) function c_fesom_gpu_profile_step(n, ms) bind(C, name='fesom_gpu_profile_step')
       import
       integer(c_int), value :: n
       real(c_double), intent(out) :: ms(7)
     end function
     integer(c_int) function c_fesom_gpu_finalize() bind(C, name='fesom_gpu_finalize')
       import
     end function
     type(c_ptr) function c_fesom_gpu_last_error() bind(C, name='fesom_gpu_last_error')
       import
     end function
     integer(c_size_t) function c_strlen(s) bind(C, name='strlen')
       import
       type(c_ptr), value :: s
     end function
  end interface

  type(fesom_part_desc), target, save :: gpart
  type(fesom_transport), save :: transport
  real(kind=WP), allocatable, target, save :: hsend(:), hrecv(:)      ! host staging of the packed halo messages
  real(kind=WP), target, save :: hred(8)
  logical, save :: is_setup = .false.
  logical, save :: builtin_transport = .false.   ! FESOM_GPU_TRANSPORT=rccl: halo exchange + solver sums by the library itself (RCCL over xGMI)

contains

  ! address of a (contiguous) module array; the reference declares its arrays without TARGET, sequence association hands
  ! over the base address without a copy
  type(c_ptr) function ar(a)
    real(kind=WP), target, intent(in) :: a(*)
    ar = c_loc(a)
  end function
  type(c_ptr) function ai(a)
    integer, target, intent(in) :: a(*)
    ai = c_loc(a)
  end function
  integer(c_int) function l2i(l)
    logical, intent(in) :: l
    l2i = merge(1_c_int, 0_c_int, l)
  end function

  subroutine check(rc, what)
    integer(c_int), intent(in) :: rc
    character(*), intent(in) :: what
    type(c_ptr) :: cmsg
    character(kind=c_char), pointer :: fmsg(:)
    integer :: n, i
    character(len=512) :: msg
    if (rc == 0) return
    msg = ''
    cmsg = c_fesom_gpu_last_error()
    if (c_associated(cmsg)) then
       n = int(min(c_strlen(cmsg), int(len(msg), c_size_t)))
       call c_f_pointer(cmsg, fmsg, (/ n /))
       do i = 1, n
          msg(i:i) = fmsg(i)
       end do
    end if
    write(*,'(a,a,a,i0,a,i0,a,a)') ' fesom_gpu: ', what, ' failed on rank ', mype, ' rc=', rc, ': ', trim(msg)      ! (one line: list-directed output wraps at 80 columns)
    flush(6)          ! status_check ends in MPI_ABORT: the message must be out before
    pe_status = 1
  end subroutine

  ! a switch of the namelists that changes oce_timestep_ale and that the library does not implement: say so and stop the run through
  ! status_check (the reference's own routine stays the one to call for such a run, INTEGRATION.md)
  subroutine refuse(cond, what)
    logical, intent(in) :: cond
    character(*), intent(in) :: what
    if (.not. cond) return
    if (mype == 0) write(*,'(a,a)') ' fesom_gpu: not implemented on the GPU path: ', what
    flush(6)
    pe_status = 1
  end subroutine

  subroutine fill_com(c, f)
    type(com_struct), intent(in) :: c
    type(fesom_com_desc), intent(out) :: f
    f%rPEnum = c%rPEnum; f%sPEnum = c%sPEnum
    f%rPE = ai(c%rPE); f%rptr = ai(c%rptr); f%rlist = ai(c%rlist)
    f%sPE = ai(c%sPE); f%sptr = ai(c%sptr); f%slist = ai(c%slist)
  end subroutine

  ! every message this rank sends is one its neighbour expects, item for item (the reference's DEBUG check, check_mpi_comm in
  ! src/gen_halo_exchange.F90:25-55): an inconsistent partition stops here through status_check instead of hanging in the first exchange
  subroutine check_plan(c, what)
    type(com_struct), intent(in) :: c
    character(*), intent(in) :: what
    integer :: scnt(0:npes-1), rexp(0:npes-1), got(0:npes-1), p, ierr
    scnt = 0; rexp = 0
    do p = 1, c%sPEnum
       scnt(c%sPE(p)) = c%sptr(p+1) - c%sptr(p)
    end do
    do p = 1, c%rPEnum
       rexp(c%rPE(p)) = c%rptr(p+1) - c%rptr(p)
    end do
    call MPI_ALLTOALL(scnt, 1, MPI_INTEGER, got, 1, MPI_INTEGER, MPI_COMM_FESOM, ierr)
    if (any(got /= rexp)) then
       write(*,*) 'fesom_gpu: halo plan mismatch (', what, ') on rank ', mype
       flush(6)
       pe_status = 1
    end if
  end subroutine

  subroutine state_desc(mesh, st)
    type(t_mesh), intent(in), target :: mesh
    type(fesom_state_desc), intent(out) :: st
    st%tr_arr = ar(tr_arr); st%tr_arr_old = ar(tr_arr_old); st%UV = ar(UV); st%UV_rhsAB = ar(UV_rhsAB)
    st%eta_n = ar(eta_n); st%d_eta = ar(d_eta); st%ssh_rhs = ar(ssh_rhs); st%ssh_rhs_old = ar(ssh_rhs_old)
    st%hbar = ar(hbar); st%hbar_old = ar(hbar_old); st%dhe = ar(dhe); st%hnode = ar(hnode); st%hnode_new = ar(hnode_new)
    st%helem = ar(helem); st%zbar_3d_n = ar(zbar_3d_n); st%Z_3d_n = ar(Z_3d_n)
    st%Wvel = ar(Wvel); st%Wvel_e = ar(Wvel_e); st%Wvel_i = ar(Wvel_i); st%ssh_values = ar(mesh%ssh_stiff%values)
  end subroutine

  subroutine fesom_gpu_setup(mesh)
    type(t_mesh), intent(in), target :: mesh
    type(fesom_mesh_desc) :: d
    type(fesom_params) :: p
    type(fesom_state_desc) :: st
    type(c_ptr) :: pp
    d%nod2D = mesh%nod2D; d%elem2D = mesh%elem2D; d%edge2D = mesh%edge2D; d%edge2D_in = mesh%edge2D_in; d%nl = mesh%nl
    d%myDim_nod2D = myDim_nod2D; d%eDim_nod2D = eDim_nod2D; d%myDim_elem2D = myDim_elem2D; d%eDim_elem2D = eDim_elem2D
    d%eXDim_elem2D = eXDim_elem2D; d%myDim_edge2D = myDim_edge2D; d%eDim_edge2D = eDim_edge2D
    d%max_nod_in_elem = size(mesh%nod_in_elem2D, 1); d%ssh_nza = size(mesh%ssh_stiff%values)
    d%myList_nod2D = ai(myList_nod2D); d%myList_elem2D = ai(myList_elem2D); d%myList_edge2D = ai(myList_edge2D)
    d%coord_nod2D = ar(mesh%coord_nod2D); d%geo_coord_nod2D = ar(mesh%geo_coord_nod2D)
    d%elem2D_nodes = ai(mesh%elem2D_nodes); d%edges = ai(mesh%edges); d%edge_tri = ai(mesh%edge_tri)
    d%elem_edges = ai(mesh%elem_edges); d%elem_neighbors = ai(mesh%elem_neighbors)
    d%nod_in_elem2D = ai(mesh%nod_in_elem2D); d%nod_in_elem2D_num = ai(mesh%nod_in_elem2D_num)
    d%nlevels = ai(mesh%nlevels); d%ulevels = ai(mesh%ulevels)
    d%nlevels_nod2D = ai(mesh%nlevels_nod2D); d%ulevels_nod2D = ai(mesh%ulevels_nod2D)
    d%nlevels_nod2D_min = ai(mesh%nlevels_nod2D_min); d%ulevels_nod2D_max = ai(mesh%ulevels_nod2D_max)
    d%zbar = ar(mesh%zbar); d%Z = ar(mesh%Z); d%depth = ar(mesh%depth); d%elem_area = ar(mesh%elem_area)
    d%area = ar(mesh%area); d%area_inv = ar(mesh%area_inv); d%areasvol = ar(mesh%areasvol); d%areasvol_inv = ar(mesh%areasvol_inv)
    d%mesh_resolution = ar(mesh%mesh_resolution)
    d%gradient_sca = ar(mesh%gradient_sca); d%gradient_vec = ar(mesh%gradient_vec)
    d%edge_dxdy = ar(mesh%edge_dxdy); d%edge_cross_dxdy = ar(mesh%edge_cross_dxdy)
    d%elem_cos = ar(mesh%elem_cos); d%metric_factor = ar(mesh%metric_factor)
    d%coriolis = ar(coriolis); d%coriolis_node = ar(coriolis_node)
    d%ssh_rowptr = ai(mesh%ssh_stiff%rowptr); d%ssh_colind = ai(mesh%ssh_stiff%colind)
    d%ssh_colind_loc = ai(mesh%ssh_stiff%colind_loc); d%ssh_values = ar(mesh%ssh_stiff%values)
    d%edge_up_dn_tri = ai(edge_up_dn_tri)
    d%zbar_n_bot = ar(zbar_n_bot); d%zbar_n_srf = ar(zbar_n_srf); d%bottom_node_thickness = ar(bottom_node_thickness)
    d%zbar_e_bot = ar(zbar_e_bot); d%zbar_e_srf = ar(zbar_e_srf); d%bottom_elem_thickness = ar(bottom_elem_thickness)

    pp = c_null_ptr
    if (npes > 1) then
       gpart%npes = npes; gpart%mype = mype
       call fill_com(com_nod2D, gpart%com_nod2D)
       call fill_com(com_elem2D, gpart%com_elem2D)
       call fill_com(com_elem2D_full, gpart%com_elem2D_full)
       pp = c_loc(gpart)
       call check_plan(com_nod2D, 'nod2D'); call check_plan(com_elem2D, 'elem2D'); call check_plan(com_elem2D_full, 'elem2D_full')
       call status_check
    end if

    call refuse(use_kpp_nonlclflx .and. mix_scheme_nmb /= 1, 'use_kpp_nonlclflx with a mixing scheme other than KPP (oce_ale_tracer.F90:725)')
    call refuse(SPP .and. .not. (allocated(thdgr) .and. allocated(S_oc_array)), 'SPP without the sea-ice arrays thdgr / S_oc_array (gen_forcing_init.F90:134, ice_setup_step.F90:127)')
    call refuse(use_momix .and. .not. allocated(mixlength), 'use_momix without the ice arrays (the reference allocates mo / mixlength only with use_ice, oce_setup_step.F90:218)')
    call status_check

    p%dt = dt
    select case (trim(which_ALE))
    case ('linfs');  p%which_ale = 0
    case ('zlevel'); p%which_ale = 1
    case default;    p%which_ale = 2        ! 'zstar'
    end select
    p%use_partial_cell = l2i(use_partial_cell); p%state_equation = state_equation; p%num_tracers = num_tracers
    p%mom_adv = mom_adv; p%visc_option = visc_option; p%i_vert_visc = l2i(i_vert_visc); p%i_vert_diff = l2i(i_vert_diff)
    if (visc_option == 8 .and. trim(which_toy) /= 'soufflet') p%visc_option = -1    ! uke_update's regional mask (oce_dyn.F90:1107-1121) is not built: the library refuses, the caller keeps the CPU step
    p%w_split = l2i(w_split)
    p%mix_scheme = mix_scheme_nmb                  ! 1 KPP, 2 PP (oce_setup_step.F90:69-82); others are rejected by the library
    p%use_instabmix = l2i(use_instabmix); p%use_windmix = l2i(use_windmix); p%windmix_nl = windmix_nl
    p%toy_soufflet = l2i(toy_ocean .and. trim(which_toy) == 'soufflet')
    p%alpha = alpha; p%theta = theta; p%epsilon = epsilon; p%C_d = C_d; p%A_ver = A_ver; p%K_ver = K_ver; p%K_hor = K_hor
    p%gamma0 = gamma0; p%gamma1 = gamma1; p%gamma2 = gamma2; p%easy_bs_return = easy_bs_return; p%w_max_cfl = w_max_cfl
    p%tra_adv_ph = tra_adv_ph; p%tra_adv_pv = tra_adv_pv; p%instabmix_kv = instabmix_kv; p%windmix_kv = windmix_kv
    p%cyclic_length = cyclic_length
    p%with_diffusion = 1; p%solver_x0_order = 3
    p%Fer_GM = l2i(Fer_GM); p%K_GM_max = K_GM_max; p%K_GM_min = K_GM_min; p%K_GM_bvref = K_GM_bvref
    p%K_GM_rampmax = K_GM_rampmax; p%K_GM_rampmin = K_GM_rampmin; p%K_GM_resscalorder = K_GM_resscalorder
    p%scaling_Ferreira = l2i(scaling_Ferreira); p%scaling_Rossby = l2i(scaling_Rossby)
    p%scaling_resolution = l2i(scaling_resolution); p%scaling_FESOM14 = l2i(scaling_FESOM14); p%Redi = l2i(Redi)
    p%visc_sh_limit = visc_sh_limit; p%diff_sh_limit = diff_sh_limit; p%Ricr = Ricr; p%concv = concv
    p%use_sw_pene = l2i(use_sw_pene)
    select case (trim(tra_adv_ver))
    case ('QR4C'); p%tra_adv_ver = 0
    case ('CDIFF'); p%tra_adv_ver = 1
    case ('UPW1'); p%tra_adv_ver = 2
    case ('PPM'); p%tra_adv_ver = 3
    case default; p%tra_adv_ver = -1
    end select
    select case (trim(tra_adv_hor))
    case ('MFCT'); p%tra_adv_hor = 0
    case ('MUSCL'); p%tra_adv_hor = 1
    case ('UPW1'); p%tra_adv_hor = 2
    case default; p%tra_adv_hor = -1
    end select
    select case (trim(tra_adv_lim))
    case ('FCT'); p%tra_adv_lim = 0
    case ('NON'); p%tra_adv_lim = 1
    case default; p%tra_adv_lim = -1
    end select
    p%Kv0_const = l2i(Kv0_const)
    p%Leith_c = Leith_c; p%Div_c = Div_c
    select case (trim(which_pgf))
    case ('shchepetkin'); p%which_pgf = 0
    case ('cubicspline'); p%which_pgf = 1
    case ('nemo'); p%which_pgf = 2
    case ('easypgf'); p%which_pgf = 3
    case ('sergey'); p%which_pgf = 4
    case default; p%which_pgf = -1
    end select
    p%use_momix = l2i(use_momix); p%momix_lat = momix_lat; p%momix_kv = momix_kv
    p%use_kpp_nonlclflx = l2i(use_kpp_nonlclflx); p%ref_sss_local = l2i(ref_sss_local); p%ref_sss = ref_sss
    p%double_diffusion = l2i(double_diffusion); p%smooth_bh_tra = l2i(smooth_bh_tra)
    p%use_floatice = l2i(use_floatice .and. .not. trim(which_ALE)=='linfs'); p%l_mslp = l2i(l_mslp); p%use_global_tides = l2i(use_global_tides)
    p%max_ice_loading = max_ice_loading; p%clim_relax = clim_relax
    p%SPP = l2i(SPP); p%Sice = Sice
    p%lzstar_lev = lzstar_lev; p%min_hnode = min_hnode
    p%c_back = c_back; p%K_back = K_back; p%uke_scaling_factor = uke_scaling_factor; p%rosb_dis = rosb_dis; p%scale_area = scale_area
    p%uke_scaling = l2i(uke_scaling); p%smooth_back = smooth_back; p%smooth_dis = smooth_dis; p%smooth_back_tend = smooth_back_tend
    p%use_cavity = l2i(use_cavity); p%use_density_ref = l2i(use_density_ref); p%density_ref_T = density_ref_T; p%density_ref_S = density_ref_S
    p%use_cavity_partial_cell = l2i(use_cavity .and. use_cavity_partial_cell)
    p%solver_precond = 1; p%solver_xinv_its = 0     ! explicit-inverse preconditioner where it fits (pi), library default iterations

    transport%ctx = c_null_ptr
    transport%exchange = c_funloc(mpi_exchange)
    transport%allreduce_sum = c_funloc(mpi_allreduce_sum)
    call check(c_fesom_gpu_init(d, pp, p), 'fesom_gpu_init')
    call status_check
    if (npes > 1) call setup_builtin_transport
    call state_desc(mesh, st)
    call check(c_fesom_gpu_upload_state(st), 'fesom_gpu_upload_state')
    ! the reference density profile ocean_setup formed from the initial layer depths (init_ref_density, oce_setup_step.F90:129): the host's array, not a recomputation from a restarted state
    if (use_density_ref .and. allocated(density_ref)) call check(c_fesom_gpu_set_field('density_ref'//c_null_char, ar(density_ref), int(size(density_ref), c_long_long)), 'fesom_gpu_set_field(density_ref)')
    if (use_momix .and. allocated(mixlength)) call check(c_fesom_gpu_set_field('mixlength'//c_null_char, ar(mixlength), int(size(mixlength), c_long_long)), 'fesom_gpu_set_field(mixlength)')
    if (clim_relax > 1.0e-8_WP .and. .not. toy_ocean) then        ! relax_to_clim: the static climatology and the nodal rate
       call check(c_fesom_gpu_set_field('Tclim'//c_null_char, ar(Tclim), int(size(Tclim), c_long_long)), 'fesom_gpu_set_field(Tclim)')
       call check(c_fesom_gpu_set_field('Sclim'//c_null_char, ar(Sclim), int(size(Sclim), c_long_long)), 'fesom_gpu_set_field(Sclim)')
       call check(c_fesom_gpu_set_field('relax2clim'//c_null_char, ar(relax2clim), int(size(relax2clim), c_long_long)), 'fesom_gpu_set_field(relax2clim)')
    end if
    call status_check
    is_setup = .true.
  end subroutine

  ! = compute_vel_nodes(mesh) + oce_timestep_ale(n, mesh) of the reference, on the GPU.  The surface forcing of this step
  ! (the arrays the reference's forcing/ice layer filled on the host) is uploaded first.
  subroutine oce_timestep_ale_gpu(n, mesh)
    integer, intent(in) :: n
    type(t_mesh), intent(in), target :: mesh
    type(fesom_forcing_desc) :: f
    real(c_double) :: pms(7)
    if (.not. is_setup) call fesom_gpu_setup(mesh)
    f%stress_surf = ar(stress_surf); f%heat_flux = ar(heat_flux); f%water_flux = ar(water_flux)
    f%virtual_salt = ar(virtual_salt); f%relax_salt = ar(relax_salt)
    f%real_salt_flux = c_null_ptr
    if (allocated(real_salt_flux)) f%real_salt_flux = ar(real_salt_flux)
    f%stress_atmoce_x = c_null_ptr; f%stress_atmoce_y = c_null_ptr
    if (allocated(stress_atmoce_x)) then
       f%stress_atmoce_x = ar(stress_atmoce_x); f%stress_atmoce_y = ar(stress_atmoce_y)
    end if
    f%sw_3d = c_null_ptr
    if (use_sw_pene .and. allocated(sw_3d)) f%sw_3d = ar(sw_3d)
    f%m_ice = c_null_ptr; f%m_snow = c_null_ptr; f%press_air = c_null_ptr; f%ssh_gp = c_null_ptr
    if (use_floatice .and. allocated(m_ice)) then
       f%m_ice = ar(m_ice); f%m_snow = ar(m_snow)
    end if
    if (l_mslp .and. allocated(press_air)) f%press_air = ar(press_air)
    if (use_global_tides .and. allocated(ssh_gp)) f%ssh_gp = ar(ssh_gp)
    f%thdgr = c_null_ptr; f%S_oc_array = c_null_ptr
    if (SPP .and. allocated(thdgr) .and. allocated(S_oc_array)) then
       f%thdgr = ar(thdgr); f%S_oc_array = ar(S_oc_array)
    end if
    f%u_ice = c_null_ptr; f%v_ice = c_null_ptr; f%a_ice = c_null_ptr
    if (use_momix .and. allocated(a_ice)) then
       f%u_ice = ar(u_ice); f%v_ice = ar(v_ice); f%a_ice = ar(a_ice)
    end if
    call check(c_fesom_gpu_set_forcing(f), 'fesom_gpu_set_forcing')
    if (fesom_gpu_profile .and. npes == 1) then
       call check(c_fesom_gpu_profile_step(int(n, c_int), pms), 'fesom_gpu_profile_step')
       rtime_oce_mixpres = rtime_oce_mixpres + pms(1)*1.0e-3_WP; rtime_oce_dyn = rtime_oce_dyn + pms(2)*1.0e-3_WP
       rtime_oce_dynssh = rtime_oce_dynssh + pms(3)*1.0e-3_WP; rtime_oce_solvessh = rtime_oce_solvessh + pms(4)*1.0e-3_WP
       rtime_oce_GMRedi = rtime_oce_GMRedi + pms(5)*1.0e-3_WP; rtime_oce_solvetra = rtime_oce_solvetra + pms(6)*1.0e-3_WP
       rtime_oce = rtime_oce + pms(7)*1.0e-3_WP
    else if (npes > 1 .and. builtin_transport) then   ! phases, halo exchange (RCCL groups) and solver sums all inside the library
       call check(c_fesom_gpu_step_partitioned_builtin(int(n, c_int), c_null_ptr), 'fesom_gpu_step_partitioned')
    else if (npes > 1) then     ! the library runs the phases and the partitioned SSH solve, this layer moves the halo bytes with MPI
       call check(c_fesom_gpu_step_partitioned(int(n, c_int), transport), 'fesom_gpu_step_partitioned')
    else
       call check(c_fesom_gpu_step(int(n, c_int)), 'fesom_gpu_step')
    end if
    call status_check
  end subroutine

  ! Built-in transport of the library (include/fesom_gpu.h): one rank per GPU, RCCL send/recv groups on the library's stream.
  ! Selected with FESOM_GPU_TRANSPORT=rccl; rank 0 draws RCCL's unique id, MPI_BCAST carries it over MPI_COMM_FESOM, every rank
  ! joins with its FESOM rank (mype), then a ring shift + global sum through the new communicator checks it end to end.
  subroutine setup_builtin_transport
    character(len=32) :: v
    character(kind=c_char) :: id(128)
    integer :: ierr, stat
    call get_environment_variable('FESOM_GPU_TRANSPORT', v, status=stat)
    builtin_transport = (stat == 0 .and. trim(v) == 'rccl')
    if (.not. builtin_transport) return
    id = c_null_char
    if (mype == 0) call check(c_fesom_gpu_comm_unique_id(id), 'fesom_gpu_comm_unique_id')
    call status_check
    call MPI_BCAST(id, 128, MPI_CHARACTER, 0, MPI_COMM_FESOM, ierr)
    call check(c_fesom_gpu_comm_init(id, int(npes, c_int), int(mype, c_int)), 'fesom_gpu_comm_init')
    call status_check
    call check(c_fesom_gpu_comm_selftest(1000_c_int), 'fesom_gpu_comm_selftest')
    call status_check
  end subroutine

  ! ---- transport callbacks of fesom_gpu_step_partitioned (include/fesom_gpu.h: fesom_transport).  Host-staged MPI: the packed
  ! device buffer is copied to the host, exchanged with MPI_Isend/Irecv along the reference's own com_struct lists
  ! (src/gen_halo_exchange.F90 does the same per field), and copied back.  With a GPU-aware MPI the two copies disappear:
  ! pass send_dev / recv_dev (converted with c_f_pointer) to MPI directly.
  integer(c_int) function mpi_exchange(ctx, kind, send_dev, recv_dev, values_per_item) bind(C)
    type(c_ptr), value :: ctx, send_dev, recv_dev
    integer(c_int), value :: kind, values_per_item
    integer :: W, ns, nr, p, first, cnt, nreq, ierr
    integer :: req(2*(npes+1))
    W = values_per_item
    mpi_exchange = 0
    select case (kind)
    case (0); call do_exchange(com_nod2D)
    case (1); call do_exchange(com_elem2D)
    case default; call do_exchange(com_elem2D_full)
    end select
  contains
    subroutine do_exchange(c)
      type(com_struct), intent(in) :: c
      ns = (c%sptr(c%sPEnum+1) - 1) * W
      nr = (c%rptr(c%rPEnum+1) - 1) * W
      if (.not. allocated(hsend)) allocate(hsend(max(ns,1)), hrecv(max(nr,1)))
      if (size(hsend) < ns) then
         deallocate(hsend); allocate(hsend(2*ns))
      end if
      if (size(hrecv) < nr) then
         deallocate(hrecv); allocate(hrecv(2*nr))
      end if
      if (ns > 0) then
         if (c_fesom_gpu_copy(c_loc(hsend), send_dev, int(ns, c_long_long)*8_c_long_long, 0_c_int) /= 0) mpi_exchange = 1
      end if
      nreq = 0
      do p = 1, c%rPEnum
         first = (c%rptr(p) - 1) * W + 1; cnt = (c%rptr(p+1) - c%rptr(p)) * W
         nreq = nreq + 1
         call MPI_IRECV(hrecv(first), cnt, MPI_DOUBLE_PRECISION, c%rPE(p), 100 + kind, MPI_COMM_FESOM, req(nreq), ierr)
      end do
      do p = 1, c%sPEnum
         first = (c%sptr(p) - 1) * W + 1; cnt = (c%sptr(p+1) - c%sptr(p)) * W
         nreq = nreq + 1
         call MPI_ISEND(hsend(first), cnt, MPI_DOUBLE_PRECISION, c%sPE(p), 100 + kind, MPI_COMM_FESOM, req(nreq), ierr)
      end do
      if (nreq > 0) call MPI_WAITALL(nreq, req, MPI_STATUSES_IGNORE, ierr)
      if (nr > 0) then
         if (c_fesom_gpu_copy(recv_dev, c_loc(hrecv), int(nr, c_long_long)*8_c_long_long, 1_c_int) /= 0) mpi_exchange = 1
      end if
    end subroutine
  end function

  integer(c_int) function mpi_allreduce_sum(ctx, buf_dev, n) bind(C)
    type(c_ptr), value :: ctx, buf_dev
    integer(c_int), value :: n
    real(kind=WP), allocatable, target, save :: loc(:), glo(:)      ! (<= 4 values in the solver, 100*(nl-1) in the Soufflet zonal means)
    integer :: ierr
    if (.not. allocated(loc)) allocate(loc(max(n, 8)), glo(max(n, 8)))
    if (size(loc) < n) then
       deallocate(loc, glo); allocate(loc(n), glo(n))
    end if
    mpi_allreduce_sum = c_fesom_gpu_copy(c_loc(loc), buf_dev, int(n, c_long_long)*8_c_long_long, 0_c_int)
    call MPI_ALLREDUCE(loc, glo, n, MPI_DOUBLE_PRECISION, MPI_SUM, MPI_COMM_FESOM, ierr)
    if (c_fesom_gpu_copy(buf_dev, c_loc(glo), int(n, c_long_long)*8_c_long_long, 1_c_int) /= 0) mpi_allreduce_sum = 1
  end function

  subroutine fesom_gpu_fetch_state(mesh)
    type(t_mesh), intent(in), target :: mesh
    type(fesom_state_desc) :: st
    call state_desc(mesh, st)
    call check(c_fesom_gpu_download_state(st), 'fesom_gpu_download_state')
    if (use_momix .and. allocated(mixlength)) call check(c_fesom_gpu_get_field('mixlength'//c_null_char, ar(mixlength), int(size(mixlength), c_long_long)), 'fesom_gpu_get_field(mixlength)')
    call status_check
  end subroutine

  subroutine fesom_gpu_push_state(mesh)       ! after the host changed the fields (restart read, nudging)
    type(t_mesh), intent(in), target :: mesh
    type(fesom_state_desc) :: st
    call state_desc(mesh, st)
    call check(c_fesom_gpu_upload_state(st), 'fesom_gpu_upload_state')
    if (use_momix .and. allocated(mixlength)) call check(c_fesom_gpu_set_field('mixlength'//c_null_char, ar(mixlength), int(size(mixlength), c_long_long)), 'fesom_gpu_set_field(mixlength)')
    call status_check
  end subroutine

  subroutine fesom_gpu_shutdown()
    integer(c_int) :: rc
    if (is_setup) rc = c_fesom_gpu_finalize()
    is_setup = .false.
  end subroutine
end module fesom_gpu_shim
