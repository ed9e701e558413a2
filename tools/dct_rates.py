import sys, time, numpy as np
sys.path.insert(0,'/root/repo/zaf-python_amd')
import zafx
for n,rows,t,sine in ((1024,16384,2,False),(1024,16384,3,False),(1024,16384,4,False),(1025,16384,1,False),(1024,16384,2,True),(4096,16384,2,False),(16384,4096,2,False),(64,262144,2,False),(1024,262144,2,False),
    (1000,16384,2,False),(1000,16384,4,True),(1001,16384,1,False),(441,65536,2,False),(100,262144,3,False),(3000,8192,2,False),(8000,2048,4,False)):   # the last rows: chirp-z lengths (k_dct_bs32)
    plan=zafx.dct_plan(n,t,sine)
    x=np.random.default_rng(0).standard_normal((rows,n)).astype(np.float32)
    d_in=zafx.DeviceBuffer.from_host(x); d_out=zafx.DeviceBuffer((rows,n),np.float32)
    for _ in range(20): plan.execute(d_in,d_out,rows,n)
    plan.sync()
    plan.timer_start()
    for _ in range(50): plan.execute(d_in,d_out,rows,n)
    ms=plan.timer_stop()/50
    print(f"n={n} rows={rows} type={t} sine={sine}: {ms*1e3:.1f} us, {rows*n*8/ms/1e6:.0f} GB/s ({plan.last_kernel})", flush=True)
    d_in.free(); d_out.free()
