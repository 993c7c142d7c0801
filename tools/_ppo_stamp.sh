# instrument k_ppo_minibatch with clock64 stamps at every barrier (block 0), run one update, restore the source
set -e
cd /root/repo
cp simple_distributed_rl_amd/csrc/srlx_ppo_net.hip /tmp/ppo_net_backup.hip
python - <<'PY'
p='/root/repo/simple_distributed_rl_amd/csrc/srlx_ppo_net.hip'
s=open(p).read()
a=s.index('__global__ void __launch_bounds__(256) k_ppo_minibatch(MbArgs a) {')
b=s.index('// partial[w][p] -> grad[p]: four lanes per parameter')
body=s[a:b]
body=body.replace('__syncthreads();', '__syncthreads();\n        if (blockIdx.x == 0 && tid == 0 && st_n < 40) st[st_n++] = clock64();')
body=body.replace('    const NetOff o = net_off(obs, A);\n','    const NetOff o = net_off(obs, A);\n    __shared__ long long st[40]; int st_n = 0; if (tid == 0) st[st_n++] = clock64();\n',1)
body=body.replace("    float *out = a.partials + (i64)blockIdx.x * a.stride;\n","    float *out = a.partials + (i64)blockIdx.x * a.stride;\n    if (blockIdx.x == 0 && tid == 0) { st[st_n++] = clock64(); for (int q = 1; q < st_n; q++) printf(\"%d:%lld \", q, st[q] - st[q - 1]); printf(\"\\n\"); }\n",1)
s=s[:a]+body+s[b:]
open(p,'w').write(s)
PY
(cd simple_distributed_rl_amd/csrc && make 2>&1 | grep -i "error" -A5 | head)
/usr/local/graft/bin/gpurun --timeout 600 -- 'python tools/_ppo_one.py 2>&1 | tail -2' 2>&1 | tail -2
cp /tmp/ppo_net_backup.hip simple_distributed_rl_amd/csrc/srlx_ppo_net.hip
(cd simple_distributed_rl_amd/csrc && make 2>&1 | grep -i "error" -A5 | head)
