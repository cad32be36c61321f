import sys, time, numpy as np
sys.path.insert(0,'/root/repo')
from frank_amd import FrankFitter, FixedGeometry
from frank_amd.mock import MOCK_GEOMETRY
from frank_amd.constants import rad_to_arcsec
import os
only = int(sys.argv[1]) if len(sys.argv) > 1 else 0
for name,N in (('sweep_N50_2e4.npz',50),('fit_N100_1e5.npz',100),('fit_N300_1e6.npz',300)):
    if only and N != only: continue
    g=np.load('/root/repo/tests/golden/'+name)
    kw={}
    if N==50: kw=dict(alpha=float(g['alpha_a']),weights_smooth=float(g['wsmooth_a']))
    FF=FrankFitter(2.0,N,FixedGeometry(**MOCK_GEOMETRY),verbose=False,store_iteration_diagnostics=True,**kw)
    m={'M':g['M'],'j':g['j'],'null_likelihood':0.0,'hash':[False,FF._DHT,FF._geometry,'opt_thick',None]}
    t=time.time(); sol=FF.fit_preprocessed(m); dt=time.time()-t
    t=time.time(); sol=FF.fit_preprocessed(m); dt=time.time()-t
    I=g['I_a'] if N==50 else g['I']; ni=int(g['niter_a'] if N==50 else g['niter'])
    nit=FF.iteration_diagnostics['num_iterations']
    print(N,'niter',nit,ni,'rel',np.abs(sol.I-I).max()/np.abs(I).max(),'time %.1f ms  %.1f us/iter'%(1e3*dt,1e6*dt/nit))

if "timing" in os.environ.get("FRANK_AMD_LIB", ""):
    import ctypes
    from frank_amd import _lib
    out=(ctypes.c_longlong*16)()
    _lib.lib.fh_debug_loop_timing(FF._DHT.context(), out)
    names=['pinv','diag factor','panel trsm','trailing','(unused)','lists (panel phase)','stage row loads','m, tr2 reduce']
    tot=sum(out[:8])

    print('chain wave: load + update %.1f, factor and invert %.1f, store + flag %.1f | worker wave1: trailing %.1f, inverse row %.1f us/iter'%tuple(v/2.35e3/(nit+2) for v in out[8:13]))
    print('outer loop: solve_posterior %.1f, beta/convergence/exp %.1f, banded solve (thread 0) %.1f us/iter' % tuple(v/2.1e3/(2*nit+4) for v in out[13:16]))
    for n_,v in zip(names,out[:8]): print('%-14s %8.1f us/iter  %5.1f%%'%(n_, v/2.1e3/ (2*nit+4) , 100*v/tot))

    nw=12
    tr=(ctypes.c_longlong*2048)()
    _lib.lib.fh_debug_loop_trace(FF._DHT.context(), tr)
    t=np.array(tr[:nw*120],dtype=np.int64).reshape(nw,20,6)
    cw=int(np.argmax((t[:,:,5]>0).sum(axis=1)))  # the wave that runs the chain
    t0=t[:,:,0][t[:,:,0]>0].min()
    us=lambda v: (v-t0)/2.4e3
    print('step | start (w0) | w0: flag set, done | workers: trailing done (min..max), flag seen (max), column done (max), all done (max)')
    for k in range(19):
        w=np.delete(t,cw,axis=0)[:,k,:]
        f=lambda a: us(a[a>0]).max() if (a>0).any() else float('nan')
        g=lambda a: us(a[a>0]).min() if (a>0).any() else float('nan')
        print('%2d  %7.2f | %7.2f %7.2f | %7.2f..%7.2f  %7.2f  %7.2f  %7.2f' % (k, us(t[cw,k,0]), us(t[cw,k,5]) if t[cw,k,5]>0 else float('nan'), us(t[cw,k,4]), g(w[:,1]), f(w[:,1]), f(w[:,2]), f(w[:,3]), f(w[:,4])))
