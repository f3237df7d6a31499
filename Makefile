# Top-level build: libb3m.so (HIP kernels + C ABI), the CLI programs, and the oracle (tests only).
HIPCC      ?= /opt/rocm/bin/hipcc
ARCH       ?= gfx950
HIPFLAGS   ?= -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -Wall -Wno-unused-function
CXXFLAGS_H ?= -O2 -std=c++17 -fPIC -Wall -pthread
CSRC       := bwtb3m_amd/csrc
OBJDIR     := build
LIB        := bwtb3m_amd/libb3m.so

HIP_SRCS   := $(CSRC)/k_scan_radix.hip $(CSRC)/k_blocksort.hip $(CSRC)/k_msd.hip $(CSRC)/k_merge.hip $(CSRC)/k_rlenc.hip $(CSRC)/engine.hip
HOST_SRCS  := $(wildcard $(CSRC)/host/*.cpp)
HIP_OBJS   := $(patsubst $(CSRC)/%.hip,$(OBJDIR)/%.o,$(HIP_SRCS))
HOST_OBJS  := $(patsubst $(CSRC)/host/%.cpp,$(OBJDIR)/host_%.o,$(filter-out $(CSRC)/host/cli_%.cpp,$(HOST_SRCS)))
HDRS       := $(wildcard $(CSRC)/*.h) $(wildcard $(CSRC)/host/*.h) include/b3m.h

all: $(LIB) bin/bwtb3m bin/bwtb3mdecoderl bin/bwtb3mtobwa bin/sortPreIsa bin/bwtdecodeblock bin/checkbwt bin/bwtcomputessa oracle

$(OBJDIR)/%.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(OBJDIR)/host_%.o: $(CSRC)/host/%.cpp $(HDRS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(CXXFLAGS_H) -x c++ -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -c $< -o $@

$(LIB): $(HIP_OBJS) $(HOST_OBJS)
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) -o $@ $^ -pthread

bin/%: $(CSRC)/host/cli_%.cpp $(LIB) $(HDRS)
	@mkdir -p bin
	g++ $(CXXFLAGS_H) -Iinclude $< -o $@ -Lbwtb3m_amd -lb3m -Wl,-rpath,'$$ORIGIN/../bwtb3m_amd' -Wl,-rpath,/opt/rocm/lib

oracle:
	$(MAKE) -C oracle -s

# debug build: every LDS index of the finish kernel is checked against its array and traps (B3M_LIB=bwtb3m_amd/libb3m_dbg.so)
DBG_OBJS := $(patsubst $(CSRC)/%.hip,$(OBJDIR)/dbg_%.o,$(HIP_SRCS))
$(OBJDIR)/dbg_%.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -DB3M_CHECK_LDS -c $< -o $@
debug: $(DBG_OBJS) $(HOST_OBJS)
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) -o bwtb3m_amd/libb3m_dbg.so $^ -pthread

# experiment build: k_finish_v3 can be left after each of its phases (B3M_FINISH_EXPERIMENT=1..5, timing only): tools/finish_experiment.py loads it with B3M_LIB
EXP_OBJS := $(patsubst $(CSRC)/%.hip,$(OBJDIR)/exp_%.o,$(HIP_SRCS))
$(OBJDIR)/exp_%.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -DB3M_PHASE_EXPERIMENT -c $< -o $@
$(OBJDIR)/exp_host_switches.o: $(CSRC)/host/switches.cpp $(HDRS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(CXXFLAGS_H) -x c++ -D__HIP_PLATFORM_AMD__ -DB3M_PHASE_EXPERIMENT -I/opt/rocm/include -c $< -o $@
experiment: $(EXP_OBJS) $(filter-out $(OBJDIR)/host_switches.o,$(HOST_OBJS)) $(OBJDIR)/exp_host_switches.o
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) -o bwtb3m_amd/libb3m_exp.so $^ -pthread

clean:
	rm -rf $(OBJDIR) $(LIB) bin
	$(MAKE) -C oracle clean

.PHONY: all oracle clean debug experiment
