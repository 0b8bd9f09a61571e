# Builds the UNMODIFIED reference (hasindu2008/cornetto v0.2.0) from the sources where they lie under
# $(REF) into oracle/_ref/ (git-ignored, travels with gpurun).  No reference source is copied into the
# repository; the reference's own Makefile is not used.  Test infrastructure only: the resulting binary /
# shared object is the checker for the oracle restatement and the "reference" CPU baseline of bench.py.
#
#   make -f oracle/ref.mk            (from the repo root; needs /root/reference, gcc, zlib)
REF     ?= /root/reference
OUT     := oracle/_ref
CC      ?= gcc
# same flags as the reference's Makefile:2 (-g dropped)
CFLAGS  := -Wall -O2 -std=c99 -w
LIBS    := -lz -lm -lpthread

SRC_TOP := main cornetto depth_main fixasm boringbits_main bigenough_main thread misc misc_p error pafrec \
           find_telomere telomere_windows telomere_breaks assbed seq asmstats nx report telocontigs
SRC_SUB := minidot/dotter minidot/paf minidot/sdict sdust/sdust
SRCS    := $(addprefix $(REF)/src/,$(addsuffix .c,$(SRC_TOP) $(SRC_SUB)))
LIBSRCS := $(filter-out $(REF)/src/main.c,$(SRCS))

all: $(OUT)/cornetto $(OUT)/libcornetto_ref.so

$(OUT)/cornetto: $(SRCS)
	@mkdir -p $(OUT)
	$(CC) $(CFLAGS) $(SRCS) $(LIBS) -o $@

# every non-static per-contig function of the reference (find, rc, process_scaffold, sdust, sdust_core,
# get_regs ...) is link-visible here, for function-level differential tests through ctypes
$(OUT)/libcornetto_ref.so: $(LIBSRCS)
	@mkdir -p $(OUT)
	$(CC) $(CFLAGS) -fPIC -shared $(LIBSRCS) $(LIBS) -o $@

clean:
	rm -rf $(OUT)
.PHONY: all clean
