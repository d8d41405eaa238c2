# Build of the product library without Python (the same commands as fesom2_amd/build.py): hipcc for the gfx950 kernels and the
# C-ABI layer, g++ for the host mesh code (see build.py for why).  Output: fesom2_amd/libfesom_gpu.so
HIPCC ?= /opt/rocm/bin/hipcc
CXX   ?= g++
SRC    = fesom2_amd/csrc
OBJ    = fesom2_amd/build
HFLAGS = --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -Wno-unused-result
HIPSRC = kernels_dyn kernels_tra kernels_toy kernels_gm kernels_kpp kernels_mon kernels_ice solver solver_ras api
OBJS   = $(addprefix $(OBJ)/,$(addsuffix .o,$(HIPSRC))) $(OBJ)/mesh_host.o $(OBJ)/precond_host.o

fesom2_amd/libfesom_gpu.so: $(OBJS)
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC -o $@ $(OBJS) -lpthread

$(OBJ)/%.o: $(SRC)/%.hip $(SRC)/dev.h $(SRC)/solver_dev.h $(SRC)/ras_host.h include/fesom_gpu.h | $(OBJ)
	$(HIPCC) $(HFLAGS) -x hip -c $< -o $@

$(OBJ)/mesh_host.o: $(SRC)/mesh_host.cpp include/fesom_gpu.h | $(OBJ)
	$(CXX) -O2 -fPIC -std=c++17 -ffp-contract=off -fno-fast-math -c $< -o $@

$(OBJ)/precond_host.o: $(SRC)/precond_host.cpp $(SRC)/ras_host.h | $(OBJ)
	$(CXX) -O3 -pthread -fPIC -std=c++17 -ffp-contract=off -fno-fast-math -c $< -o $@

$(OBJ):
	mkdir -p $(OBJ)

# Fortran host layer, compiled against a FESOM2 build's module files (FESOM_MOD = directory with o_param.mod, mod_mesh.mod, ...)
FC ?= /opt/rocm/bin/amdflang
fesom_gpu_shim.o: fesom2_amd/fortran/fesom_gpu_shim.F90
	$(FC) -cpp -fdefault-real-8 -O2 -I$(FESOM_MOD) -I/opt/conda/include -c $< -o $@

clean:
	rm -rf $(OBJ) fesom2_amd/libfesom_gpu.so fesom_gpu_shim.o
.PHONY: clean
