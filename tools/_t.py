import sys,time; sys.path.insert(0,'.')
import numpy as np
from cudaimageprocessing_amd import ops, synth as S
for (w,h) in ((675,900),(2592,2000),(1024,1024),(4096,4096)):
    img=S.pattern_image("natural",w,h,3); out=np.empty_like(img)
    ops.imageFFTImageGenerate(out,img,w,h)
    t=time.perf_counter()
    for _ in range(5): ops.imageFFTImageGenerate(out,img,w,h)
    dt=(time.perf_counter()-t)/5
    t=time.perf_counter()
    for _ in range(5): ops.compressImage(out,img,w,h,0.1)
    dc=(time.perf_counter()-t)/5
    print(w,h,f"fft_highpass {dt*1e3:.2f} ms ({w*h/dt/1e6:.0f} MP/s)  compress {dc*1e3:.2f} ms",flush=True)
