"""Build an instrumented copy of conv_igemm.hip (cycle-counter timestamps per phase, thread 0 of each block)."""
import subprocess, os
src=open('/root/repo/robustart_amd/csrc/conv_igemm.hip').read()
s=src
def rep(old,new,cnt=1):
    global s
    assert old in s, old[:60]
    s=s.replace(old,new,cnt)
rep('namespace {\nconstexpr int BM = 128;','__device__ unsigned long long g_dbg[8192 * 16];\nextern "C" int rart_dbg_read(unsigned long long* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_dbg), sizeof(g_dbg)); }\n#define TS(k) if (tid == 0 && blockIdx.x < 8192) g_dbg[blockIdx.x * 16 + (k)] = __builtin_readcyclecounter();\nnamespace {\nconstexpr int BM = 128;')
rep('  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;','  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;\n  TS(0)')
rep('''  RART_LOAD_TILE(0, 0);
  if (KT > 1) RART_LOAD_TILE(1, 1);
  RART_STORE_TILE(0, 0);
  __syncthreads();''','''  RART_LOAD_TILE(0, 0);
  if (KT > 1) RART_LOAD_TILE(1, 1);
  TS(1)
  RART_STORE_TILE(0, 0);
  __syncthreads();
  TS(2)''')
rep('''    RART_LOAD64(0);
    RART_STORE64(0);
    __syncthreads();''','''    RART_LOAD64(0);
    TS(1)
    RART_STORE64(0);
    __syncthreads();
    TS(2)''')
rep('  // ---- epilogue: each wave transposes','  TS(3)\n  // ---- epilogue: each wave transposes')
rep('''    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");''','''    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (i == 0) { TS(6) } else { TS(8) }''')
rep('''    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}
}  // namespace''','''    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (i == 0) { TS(7) }
  }
  TS(4)
  __builtin_amdgcn_s_waitcnt(0);
  TS(5)
}
}  // namespace''')
p='/root/repo/scratch/exp/conv_igemm_ts.hip'
open(p,'w').write(s)
objs=[os.path.join('/root/repo/robustart_amd/csrc/_obj',f) for f in os.listdir('/root/repo/robustart_amd/csrc/_obj') if f.endswith('.o') and not f.startswith('conv_igemm')]
o='/root/repo/scratch/exp/ci_ts.o'
r=subprocess.run(['/opt/rocm/bin/hipcc','--offload-arch=gfx950','-O3','-std=c++17','-fPIC','-I','/root/repo/include','-I','/root/repo/robustart_amd/csrc','-c',p,'-o',o],capture_output=True,text=True)
assert r.returncode==0, r.stderr[-3000:]
os.makedirs('/root/repo/scratch/exp/ts',exist_ok=True)
r=subprocess.run(['/opt/rocm/bin/hipcc','--offload-arch=gfx950','-shared','-fPIC','-o','/root/repo/scratch/exp/ts/librobustart_hip.so']+objs+[o],capture_output=True,text=True)
assert r.returncode==0, r.stderr[-3000:]
print('ts lib ok')
