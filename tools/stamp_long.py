"""Where a wave of k_long spends its cycles, phase by phase (development aid; needs a GPU and a library built with
-DAMP_WV_STAMPS: the stamps add s_memtime reads to wave_read and eight atomics per wave at the end of the kernel).
usage: stamp_long.py"""
import sys; sys.argv=['x','10']
exec(open('tools/time_longreads.py').read().split("for name, trim, count")[0])
e.reset(); e.process(b, want_trim=False)
dc = e.debug_counters()
names=['load+classify','primer clips','qwindow+scan','quality clip','final pass+outputs','match bases','indel lanes','-']
tot=sum(int(dc[8+k]) for k in range(7))
for k in range(7): print('%-20s %8.2f us per read (shader clock / 100) %5.1f %%'%(names[k], int(dc[8+k])/b.n/100.0, 100.0*int(dc[8+k])/tot))
print('total per read', tot/b.n/100.0,'us')
