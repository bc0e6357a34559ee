# Same targets as the reference's Makefile (/root/reference/Makefile:11-21), built for gfx950.
#   make            library + driver      (reference: `make cbet-gpu`, nvcc sm_70)
#   make test       ./cbet-gpu 10 in PRINT mode, compared with the golden text   (reference :14-17)
#   make clean
PY      ?= python
LIBDIR   = cbet_raytracing_3d_amd/lib
# md5 / size of the 100^3 text dump recorded by the survey's host compile of the reference kernel (SURVEY.md 8(c),
# tests/test_oracle_golden.py).  The reference's golden file truth_100 is not in the mount (.MISSING_LARGE_BLOBS): this
# digest has NEVER been compared with it.  When a truth_100 is present in this directory, `make test` cmp's against it.
TRUTH_100_MD5   = cc0909ed1c5938704c51165dc20cb829
TRUTH_100_BYTES = 12544620

all: cbet-gpu

cbet-gpu: $(LIBDIR)/cbet-gpu

$(LIBDIR)/cbet-gpu: cbet_raytracing_3d_amd/csrc/*.hip cbet_raytracing_3d_amd/csrc/*.cpp cbet_raytracing_3d_amd/csrc/*.h include/*.h include/*.hpp tools/cbet_gpu.cpp
	$(PY) -m cbet_raytracing_3d_amd.build --force

test: cbet-gpu
	$(LIBDIR)/cbet-gpu 10 --n 100 --print > cbet_gpu_output
	@if [ -f truth_100 ]; then cmp cbet_gpu_output truth_100 && echo "PASS: identical to truth_100"; \
	else test "$$(wc -c < cbet_gpu_output)" = "$(TRUTH_100_BYTES)" && \
	     test "$$(md5sum < cbet_gpu_output | cut -d' ' -f1)" = "$(TRUTH_100_MD5)" && \
	     echo "PASS: $(TRUTH_100_BYTES) bytes, md5 $(TRUTH_100_MD5) (digest recorded by the survey's host compile of the reference kernel; never compared with truth_100, which is absent)"; fi

clean:
	$(RM) $(LIBDIR)/cbet-gpu $(LIBDIR)/cbet-ref-shaped $(LIBDIR)/libcbet_mi355x.so cbet_gpu_output oracle/libcbet_oracle.so

.PHONY: all cbet-gpu test clean
