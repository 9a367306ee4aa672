"""Per-CU timeline of a phase log of the hierarchical kernel (ICP_NN_PHASES=file, large models): wave 2 of every block
leaves XCC_ID << 32 | HW_ID in its slot 6; prints the duration of the blocks and the gap between the end of one block and the
start of the next on the same CU.   usage: python tools/cu_gaps.py ph.bin"""
import sys, numpy as np
a = np.fromfile(sys.argv[1], dtype=np.int64); a = a[:len(a)//10*10].reshape(-1,10)
nb = len(a)//16
a = a[:nb*16].reshape(nb,16,10)
live = a[:,0,1] > 0
a = a[live]; nb=len(a)
start = a[:,:,0].min(1); end = a[:,0,9]
hw = a[:,2,6]; xcc = (hw>>32)&0xf; hwid = hw & 0xffffffff
cu = (hwid>>8)&0xf; sh=(hwid>>12)&1; se=(hwid>>13)&7
key = xcc*1000 + se*100 + sh*20 + cu
print("blocks", nb, "distinct CUs", len(np.unique(key)), "xcc", np.unique(xcc), "se", np.unique(se), "sh", np.unique(sh), "cu", np.unique(cu))
gaps=[]; durs=[]; conc=[]
for k in np.unique(key):
    m = key==k; s_=start[m]; e_=end[m]; o=np.argsort(s_); s_=s_[o]; e_=e_[o]
    durs += list((e_-s_)/100.0)
    gaps += list((s_[1:]-e_[:-1])/100.0)
gaps=np.array(gaps); durs=np.array(durs)
print("block duration us: median %.2f mean %.2f" % (np.median(durs), durs.mean()))
print("gap end->next start on the same CU us: median %.2f mean %.2f p10 %.2f p90 %.2f min %.2f  (negative = overlap)" % (np.median(gaps), gaps.mean(), np.percentile(gaps,10), np.percentile(gaps,90), gaps.min()))
print("blocks per CU: ", np.bincount(np.unique(key, return_inverse=True)[1]).min(), np.bincount(np.unique(key, return_inverse=True)[1]).max())
