"""gpurun_out/prof_r01 (scratch/collect_profiles.sh) -> profiles/r01_* summaries."""
import pandas as pd, glob, json, shutil
R='gpurun_out/prof_r01'
ks=pd.read_csv(sorted(glob.glob(R+'/stats/runc/*kernel_stats.csv'))[-1])
ks2=ks.copy(); ks2['Name']=ks2['Name'].str.slice(0,140)
ks2.to_csv('profiles/r01_kernel_stats.csv', index=False)
kt=pd.read_csv(sorted(glob.glob(R+'/stats/runc/*kernel_trace.csv'))[-1])
kt['dur']=kt.End_Timestamp-kt.Start_Timestamp
sol=kt[kt.Kernel_Name.str.contains('rp_stage_kernel<double, 1',regex=False)]
pos=kt[kt.Kernel_Name.str.contains('rp_stage_kernel<double, 0',regex=False)]
posfull=pos[pos.dur>50000]
task=kt[kt.Kernel_Name.str.contains('rp_task_',regex=False)]
other=kt[~kt.Kernel_Name.str.contains('rp_stage_kernel|rp_task_|rp_reset',regex=True)]
nstep=max(1,len(task))
cfg=lambda df:{k:str(df.iloc[0][k]) for k in ['LDS_Block_Size','Scratch_Size','VGPR_Count','Accum_VGPR_Count','SGPR_Count'] if k in df.columns}
out={"round":1,
 "command":"rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --no-cpu-baseline --aux-fp32 0 --host-io 0  (fp64 engine, 4096 envs, full env.step, 20 warm-up + 158 timed steps)",
 "kernels":{
  "rp_stage_kernel<double, 1, 4> (solver stage, dominant)":{"launches":int(len(sol)),"avg_us":float(sol.dur.mean()/1e3),"min_us":float(sol.dur.min()/1e3),"max_us":float(sol.dur.max()/1e3),"share_of_gpu_time":float(sol.dur.sum()/kt.dur.sum()),"launch_config":cfg(sol)},
  "rp_stage_kernel<double, 0, 0> (position/velocity stage)":{"launches":int(len(posfull)),"avg_us":float(posfull.dur.mean()/1e3),"masked_forward_launches":int(len(pos)-len(posfull)),"share_of_gpu_time":float(pos.dur.sum()/kt.dur.sum()),"launch_config":cfg(posfull)},
  "rp_task_advance_kernel<double> (fused task layer)":{"launches":int(len(task)),"avg_us":float(task.dur.mean()/1e3) if len(task) else None,"share_of_gpu_time":float(task.dur.sum()/kt.dur.sum())},
  "torch kernels (action scaling, ctrl scatter, masks, output copies)":{"launches_per_step":float(len(other)/nstep),"share_of_gpu_time":float(other.dur.sum()/kt.dur.sum())}},
}
pm={}
for name,d in (('FETCH_SIZE','fetch'),('WRITE_SIZE','write')):
    df=pd.read_csv(sorted(glob.glob(f'{R}/{d}/runc/*counter_collection.csv'))[-1])
    df=df[df.Counter_Name==name]
    for tag,pat in (('solver','<double, 1'),('position','<double, 0')):
        x=df[df.Kernel_Name.str.contains(pat,regex=False)]
        if tag=='position': x=x[x.Counter_Value>x.Counter_Value.max()*0.05]
        pm[f'{name}_KB_per_launch_{tag}']=float(x.Counter_Value.mean()); pm[f'n_{name}_{tag}']=int(len(x))
sol_bytes = pm['FETCH_SIZE_KB_per_launch_solver']*1024*2 + pm['WRITE_SIZE_KB_per_launch_solver']*1024
pos_bytes = pm['FETCH_SIZE_KB_per_launch_position']*1024*2 + pm['WRITE_SIZE_KB_per_launch_position']*1024
pm['note']="separate --pmc passes (5 warm-up + 20 timed steps each, bench.py --steps 20 --warmup 5). Per MI355X_MICROARCH.md (HBM section) FETCH_SIZE on gfx950 counts 64 B per 128-B request, so the read side is doubled; WRITE_SIZE is used as reported (uncalibrated). These counters sit on the L2's fabric side: Infinity-Cache hits are included, so this is an upper bound on HBM traffic."
pm['solver_kernel_bytes_per_launch_corrected']=sol_bytes
pm['position_kernel_bytes_per_launch_corrected']=pos_bytes
out['pmc']=pm
json.dump(out, open('profiles/r01_step_kernel_summary.json','w'), indent=1)
json.dump({"envs":4096,"precision":64,"solver_kernel_bytes_per_launch":sol_bytes,"position_kernel_bytes_per_launch":pos_bytes}, open('profiles/traffic_r01.json','w'))
shutil.copy(R+'/bench_plain.json','profiles/r01_bench.json')
print(json.dumps(out,indent=1)[:2400])
d=json.load(open('profiles/r01_bench.json'))
print(d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms'], d['roofline']['step_sequence_avg_ms'], d['roofline']['frac'], d['aux']['fp32_engine']['value'], d['cpu_baseline']['value'], d['cpu_baseline']['sample'])
