! Oracle harness stubs (own code, NOT reference source).
! The reference modules below need NetCDF, which this image lacks; the hot path
! never calls into them for the configurations the oracle runs (no ice, no
! forcing files, no output).  Interfaces mirror what the reference `use`s:
!   g_sbf                 <- src/gen_surface_forcing.F90 (only l_mslp is read, oce_ale_vel_rhs.F90:24,67)
!   io_RESTART            <- src/io_restart.F90 (only `use`d, oce_ale.F90:1700,2528)
!   io_BLOWUP             <- src/io_blowup.F90  (blowup(istep,mesh), write_step_info.F90:228,500)
!   g_read_other_NetCDF   <- src/gen_modules_read_NetCDF.F90 (cvmix idemix/tidal only)
!   g_ic3d                <- src/gen_ic3d.F90 (namelist oce_init3d + do_ic3d, oce_setup_step.F90:478)
module g_sbf
  implicit none
  logical :: l_mslp = .false.
  logical :: l_cloud = .false.
  logical :: l_snow = .false.
end module g_sbf

module io_RESTART
  implicit none
end module io_RESTART

module io_BLOWUP
  implicit none
contains
  subroutine blowup(istep, mesh)
    use MOD_MESH
    integer :: istep
    type(t_mesh), intent(in), target :: mesh
    write(*,*) 'ORACLE: blowup() called at step', istep
  end subroutine blowup
end module io_BLOWUP

module g_read_other_NetCDF
  implicit none
contains
  subroutine read_other_NetCDF(file, vari, itime, model_2Darray, check_dummy, mesh)
    use MOD_MESH
    use g_PARSUP
    character(*), intent(in) :: file, vari
    integer :: itime
    real(kind=8) :: model_2Darray(:)
    logical :: check_dummy
    type(t_mesh), intent(in), target :: mesh
    write(*,*) 'ORACLE: read_other_NetCDF is not available (no NetCDF)'
    call par_ex(1)
  end subroutine read_other_NetCDF
end module g_read_other_NetCDF

! Initial conditions: the oracle reads T and S from raw fp64 files written by
! tests/golden/make_ic.py (global node order, (nl-1) x nod2D, column-major) so
! that the reference and the build start from bit-identical tracers.
module g_ic3d
  use o_ARRAYS
  use MOD_MESH
  use o_PARAM
  use g_PARSUP
  implicit none
  integer, parameter :: ic_max=10
  logical, save :: t_insitu =.true.
  integer, save :: n_ic3d
  integer, save, dimension(ic_max) :: idlist
  character(MAX_PATH), save, dimension(ic_max) :: filelist
  character(50), save, dimension(ic_max) :: varlist
  namelist / oce_init3d / n_ic3d, idlist, filelist, varlist, t_insitu
contains
  subroutine do_ic3d(mesh)
    type(t_mesh), intent(in), target :: mesh
    real(kind=8), allocatable :: buf(:,:)
    integer :: n, tr, u, nlm1
    character(len=16) :: fn(2)
    fn(1) = 'ic_T.bin'
    fn(2) = 'ic_S.bin'
    nlm1 = mesh%nl-1
    allocate(buf(nlm1, mesh%nod2D))
    do tr=1,2
       open(newunit=u, file=trim(fn(tr)), access='stream', form='unformatted', status='old')
       read(u) buf
       close(u)
       do n=1, myDim_nod2D+eDim_nod2D
          tr_arr(:,n,tr) = buf(:, myList_nod2D(n))
       end do
    end do
    deallocate(buf)
  end subroutine do_ic3d
end module g_ic3d
