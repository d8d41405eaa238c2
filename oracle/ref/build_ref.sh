#!/bin/bash
# Builds the REAL reference hot path (Fortran + pARMS C, sources compiled where
# they lie under /root/reference) together with this repo's harness driver and
# stub modules into oracle/_ref/fesom_oracle.x.  Test infrastructure only.
# Nothing from /root/reference is copied into the repo; outputs go to oracle/_ref/
# (git-ignored).  Needs: amdflang (ROCm), gcc, MPICH from /opt/conda.
set -e
REF=${REF:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=$(cd "$HERE/.." && pwd)/_ref
FC=${FC:-/opt/rocm/bin/amdflang}
MPI_INC=/opt/conda/include
MPI_LIB=/opt/conda/lib
mkdir -p "$OUT/obj" "$OUT/parms_obj"
if [ ! -d "$REF/src" ]; then
  echo "build_ref.sh: $REF not present (GPU box?) - keeping prebuilt oracle/_ref as is"; exit 0
fi
if [ -x "$OUT/fesom_oracle.x" ] && [ "$OUT/fesom_oracle.x" -nt "$HERE/driver.F90" ] && [ "$OUT/fesom_oracle.x" -nt "$HERE/stubs.F90" ] && [ -x "$OUT/fesom_gpu_dropin.x" ] && [ -x "$OUT/fesom_psolve_gpu.x" ] && \
   [ "$OUT/fesom_gpu_dropin.x" -nt "$HERE/../../fesom2_amd/fortran/fesom_gpu_shim.F90" ] && [ "$OUT/fesom_gpu_dropin.x" -nt "$HERE/driver.F90" ] && [ "$OUT/fesom_psolve_gpu.x" -nt "$HERE/../../fesom2_amd/fortran/fesom_gpu_psolve_mpi.c" ] && [ -z "$FORCE" ]; then
  echo "build_ref.sh: up to date"; exit 0
fi
FFLAGS="-cpp -DPARMS -fdefault-real-8 -O2 -I$MPI_INC -I$REF/src -I$REF/lib/parms/include -module-dir $OUT/obj -I$OUT/obj"

# ---- pARMS (C) ----
cd "$OUT/parms_obj"
if [ ! -f "$OUT/libparms.a" ]; then
  for f in $REF/lib/parms/src/*.c $REF/lib/parms/src/DDPQ/*.c; do
    b=$(basename $f .c)
    [ "$(dirname $f)" = "$REF/lib/parms/src/DDPQ" ] && b=ddpq_$b
    gcc -O2 -w -fPIC -DPARMS -DUSE_MPI -DREAL=double -DDBL -DFORTRAN_UNDERSCORE -DVOID_POINTER_SIZE_8 \
      -I$REF/lib/parms/include -I$REF/lib/parms/src/include -I$REF/lib/parms/src -I$REF/src -I$MPI_INC \
      -c $f -o $b.o &
    while [ $(jobs -r | wc -l) -ge 8 ]; do sleep 0.05; done
  done
  wait
  ar rcs "$OUT/libparms.a" *.o
fi
gcc -O2 -w -fPIC -DPARMS -DUSE_MPI -DREAL=double -DDBL -DFORTRAN_UNDERSCORE -DVOID_POINTER_SIZE_8 \
  -I$REF/lib/parms/include -I$REF/lib/parms/src/include -I$REF/lib/parms/src -I$REF/src -I$MPI_INC \
  -c $REF/src/psolve.c -o "$OUT/obj/psolve.o"

# ---- Fortran, dependency order ----
cd "$OUT/obj"
S=$REF/src
LIST="$S/oce_modules.F90 $S/MOD_MESH.F90 $S/gen_modules_config.F90 $S/gen_modules_partitioning.F90 $S/gen_modules_clock.F90
$S/gen_modules_rotate_grid.F90 $S/gen_halo_exchange.F90 $S/ice_modules.F90 $S/gen_modules_forcing.F90 $S/gen_support.F90
$HERE/stubs.F90
$S/oce_ale_mixing_kpp.F90 $S/oce_adv_tra_hor.F90 $S/oce_adv_tra_ver.F90 $S/oce_adv_tra_fct.F90 $S/oce_adv_tra_driver.F90
$S/gen_modules_diag.F90 $S/oce_ale_mixing_pp.F90 $S/oce_tracer_mod.F90
$S/cvmix_kinds_and_types.F90 $S/cvmix_utils.F90 $S/cvmix_put_get.F90 $S/cvmix_tke.F90 $S/cvmix_idemix.F90
$S/gen_modules_cvmix_idemix.F90 $S/gen_modules_cvmix_tke.F90 $S/cvmix_math.F90 $S/cvmix_kpp.F90 $S/gen_modules_cvmix_kpp.F90
$S/cvmix_tidal.F90 $S/gen_modules_cvmix_tidal.F90 $S/cvmix_shear.F90 $S/gen_modules_cvmix_pp.F90
$S/toy_channel_soufflet.F90 $S/gen_comm.F90 $S/oce_setup_step.F90 $S/oce_mesh.F90 $S/oce_dyn.F90 $S/oce_ale_vel_rhs.F90
$S/oce_vel_rhs_vinv.F90 $S/oce_ale_pressure_bv.F90 $S/oce_fer_gm.F90 $S/oce_muscl_adv.F90 $S/oce_ale.F90 $S/oce_ale_tracer.F90
$S/write_step_info.F90 $S/oce_mo_conv.F90 $S/oce_spp.F90 $S/cavity_param.F90 $S/ice_maEVP.F90 $S/ice_EVP.F90 $S/ice_fct.F90 $S/ice_thermo_oce.F90
$HERE/driver.F90"
OBJS=""
for f in $LIST; do
  b=$(basename $f .F90)
  if [ ! -f $b.o ] || [ $f -nt $b.o ] || [ -n "$FORCE" ]; then
    echo "FC $b"
    $FC $FFLAGS -c $f -o $b.o 2> $b.log || { cat $b.log | grep -v warning | head -40; exit 1; }
  fi
  OBJS="$OBJS $b.o"
done
$FC -O2 -o "$OUT/fesom_oracle.x" $OBJS psolve.o "$OUT/libparms.a" -L$MPI_LIB -lmpifort -lmpi -Wl,-rpath,$MPI_LIB
echo "built $OUT/fesom_oracle.x"

# ---- the same harness with the GPU drop-in: the repo's Fortran host layer (fesom2_amd/fortran/fesom_gpu_shim.F90) compiled
#      against the reference's modules and linked to libfesom_gpu.so; driver mode 'gpu' steps through it
REPO=$(cd "$HERE/../.." && pwd)
GPULIB=$REPO/fesom2_amd
if [ -f "$GPULIB/libfesom_gpu.so" ]; then
  echo "FC fesom_gpu_shim"
  $FC $FFLAGS -c $GPULIB/fortran/fesom_gpu_shim.F90 -o fesom_gpu_shim.o 2> fesom_gpu_shim.log || { grep -v warning fesom_gpu_shim.log | head -40; exit 1; }
  # the reference's oce_ale.F90 once more with its time step renamed by the preprocessor (no source edit), so that the NAME
  # oce_timestep_ale -- what fvom_main.F90:250 calls -- resolves to the repo's drop-in (fesom_gpu_oce_timestep_ale.F90);
  # module files of this variant go to their own directory
  mkdir -p gpumod
  GFLAGS="-cpp -DPARMS -fdefault-real-8 -O2 -I$MPI_INC -I$REF/src -I$REF/lib/parms/include -module-dir $OUT/obj/gpumod -I$OUT/obj"
  $FC $GFLAGS -Doce_timestep_ale=oce_timestep_ale_cpu -c $S/oce_ale.F90 -o oce_ale_cpuname.o 2> oce_ale_cpuname.log || { grep -v warning oce_ale_cpuname.log | head -40; exit 1; }
  $FC $FFLAGS -c $GPULIB/fortran/fesom_gpu_oce_timestep_ale.F90 -o fesom_gpu_oce_timestep_ale.o 2> fesom_gpu_oce_timestep_ale.log || { grep -v warning fesom_gpu_oce_timestep_ale.log | head -40; exit 1; }
  $FC $GFLAGS -DWITH_GPU_SHIM -I$OUT/obj/gpumod -c $HERE/driver.F90 -o driver_gpu.o 2> driver_gpu.log || { grep -v warning driver_gpu.log | head -40; exit 1; }
  GOBJS=$(echo $OBJS | sed 's/ driver.o/ fesom_gpu_shim.o fesom_gpu_oce_timestep_ale.o driver_gpu.o/; s/ oce_ale.o/ oce_ale_cpuname.o/')
  $FC -O2 -o "$OUT/fesom_gpu_dropin.x" $GOBJS psolve.o "$OUT/libparms.a" -L$MPI_LIB -lmpifort -lmpi -L$GPULIB -lfesom_gpu \
    -Wl,-rpath,$MPI_LIB -Wl,-rpath,/root/repo/fesom2_amd -Wl,-rpath,$GPULIB
  echo "built $OUT/fesom_gpu_dropin.x"
  # ---- the reference's CPU time step with ONLY the SSH solve replaced: same objects, no psolve.c / pARMS, the three psolve
  #      entry points (src/psolve.c:16,117,152) resolved by libfesom_gpu.so
  #      + the MPI host adapter of the solver (fesom2_amd/fortran/fesom_gpu_psolve_mpi.c): psolver_init / psolve for any number of ranks
  gcc -O2 -Wall -fPIC -I$MPI_INC -I$REPO/include -c $GPULIB/fortran/fesom_gpu_psolve_mpi.c -o fesom_gpu_psolve_mpi.o
  $FC -O2 -o "$OUT/fesom_psolve_gpu.x" $OBJS fesom_gpu_psolve_mpi.o -L$MPI_LIB -lmpifort -lmpi -L$GPULIB -lfesom_gpu \
    -Wl,-rpath,$MPI_LIB -Wl,-rpath,/root/repo/fesom2_amd -Wl,-rpath,$GPULIB
  echo "built $OUT/fesom_psolve_gpu.x"
fi
