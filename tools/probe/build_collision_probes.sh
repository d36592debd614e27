#!/bin/bash
# Development aid (not part of the product): builds two instrumented copies of the library from the current sources --
# hoic_amd/libhoic_colprof.so (tools/phase_timing.py with HOIC_LIB=libhoic_colprof.so: the collision stage split into sphere + box
# tests [slot "M^-1 solve"], per-lane routines ["diff+reward"], box-box turns ["pd_torque"], hull turns ["kinematics"], compaction
# ["collision"]) and hoic_amd/libhoic_colcnt.so (tools/probe/mesh_counts.py: event counters of the narrow phase).  The patches are
# applied to a copy under /tmp; they assert that the source lines they hook still exist.
set -e
rm -rf /tmp/colprof && mkdir -p /tmp/colprof/hoic_amd && cp -r /root/repo/hoic_amd/csrc /tmp/colprof/hoic_amd/ && cp -r /root/repo/include /tmp/colprof/
cd /tmp/colprof/hoic_amd/csrc
cp hoic_collide.h hoic_collide.h.orig; cp hoic_capi.hip hoic_capi.hip.orig
python - <<'PY'
p='hoic_collide.h'
s=open(p).read()
def rep(a,b):
    global s
    assert s.count(a)==1, a
    s=s.replace(a,b,1)
rep("    // box-box pairs: one after the other, the whole wave on each (col_box_box_wave)\n","    PT(13);\n    // box-box pairs: one after the other, the whole wave on each (col_box_box_wave)\n")
rep("    // mesh pairs: likewise one after the other","    PT(1);\n    // mesh pairs: likewise one after the other")
rep("    // margin filter, then the survivors go","    PT(3);\n    // margin filter, then the survivors go")
rep("      if (test && !isbb && !ismesh) {\n        const float s1[3]","      PT(8);\n      if (test && !isbb && !ismesh) {\n        const float s1[3]")
open(p,'w').write(s)
PY
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-value -Wno-unused-result -Wno-comment -ffp-contract=fast -mllvm -amdgpu-mfma-vgpr-form"
/opt/rocm/bin/hipcc $FL -DHOIC_BUILD_ID=\"colprof\" -DHOIC_PHASE_TIMING -c -o a.o hoic_capi.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/hoic_amd/libhoic_colprof.so a.o /root/repo/hoic_amd/csrc/hoic_mlp.o
cp hoic_collide.h.orig hoic_collide.h
python - <<'PY'
p='hoic_collide.h'
s=open(p).read()
def rep(a,b):
    global s
    assert s.count(a)==1, a
    s=s.replace(a,b,1)
rep('typedef float f4v __attribute__((ext_vector_type(4)));','typedef float f4v __attribute__((ext_vector_type(4)));\n__device__ unsigned long long g_cnt[16];\n#define CNT(i) do { if (threadIdx.x == 0) atomicAdd(&g_cnt[i], 1ull); } while (0)\n')
rep('''  float bv = -1e30f; int bi = 0x00ffffff;
  if (!h.prune || h.np <= HULL_STREAM_BELOW) {''','''  float bv = -1e30f; int bi = 0x00ffffff;
  CNT(1);
  if (!h.prune || h.np <= HULL_STREAM_BELOW) { CNT(2);''')
rep('''      while (mask) {
        const int ra = __ffsll((long long)mask) - 1;''','''      while (mask) { CNT(3);
        const int ra = __ffsll((long long)mask) - 1;''')
rep('''        int nn;
        if (ta == HOIC_GEOM_CAPSULE) nn = col_capsule_mesh_wave''','''        int nn; CNT(0); if (ta == HOIC_GEOM_CAPSULE) CNT(4); else if (ta == HOIC_GEOM_BOX) CNT(5); else CNT(6);
        if (ta == HOIC_GEOM_CAPSULE) nn = col_capsule_mesh_wave''')
rep('''        if (tid == L) lc.n = min(nn, lc.cap);
      }
      wsync();
    }
    // margin filter''','''        if (nn > 0) CNT(7);
        if (tid == L) lc.n = min(nn, lc.cap);
      }
      wsync();
    }
    // margin filter''')
rep('''      if (test && !isbb && !ismesh) {
        const float s1[3]''','''      { const unsigned long long nb = __ballot(test && !isbb && !ismesh && t1 == HOIC_GEOM_CAPSULE && t2 == HOIC_GEOM_BOX); if (nb) CNT(8);
        const unsigned long long nc = __ballot(test && !isbb && !ismesh && t1 == HOIC_GEOM_CAPSULE && t2 == HOIC_GEOM_CAPSULE); if (nc) CNT(9);
        const unsigned long long np_ = __ballot(test && !isbb && !ismesh && t1 == HOIC_GEOM_PLANE && t2 == HOIC_GEOM_BOX); if (np_) CNT(10);
        const unsigned long long nq = __ballot(test && !isbb && !ismesh && t1 == HOIC_GEOM_PLANE && t2 == HOIC_GEOM_CAPSULE); if (nq) CNT(11);
        if (threadIdx.x == 0) { atomicAdd(&g_cnt[12], (unsigned long long)__popcll(nb)); atomicAdd(&g_cnt[13], (unsigned long long)__popcll(nc)); } }
      if (test && !isbb && !ismesh) {
        const float s1[3]''')
open(p,'w').write(s)
p='hoic_capi.hip'
s=open(p).read()
s=s.replace('extern "C" int32_t hoicdbg_env_ncon(','extern "C" int32_t hoicdbg_cnt(unsigned long long* out) { hipDeviceSynchronize(); return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cnt), 128) == hipSuccess ? 0 : -1; }\nextern "C" int32_t hoicdbg_env_ncon(',1)
open(p,'w').write(s)
PY
/opt/rocm/bin/hipcc $FL -DHOIC_BUILD_ID=\"colcnt\" -c -o b.o hoic_capi.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/hoic_amd/libhoic_colcnt.so b.o /root/repo/hoic_amd/csrc/hoic_mlp.o
ls -la /root/repo/hoic_amd/libhoic_col*.so
