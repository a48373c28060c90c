# LDS bank-conflict model of the transform passes (k_ntt29_pass): extra cycles per access for candidate paddings of the
# nine-word element records, over the stage pairs of each pass shape.  bank = word mod 32, a 32-lane half at a time
# (ds_read2_b32 / ds_write_b32, MI355X_MICROARCH.md LDS table).  usage: python tools/lds_model.py
import numpy as np
NT=512
def rows_for(lbs, last=None, TS=2048, NT=512):
    rows={}
    b=np.arange(TS//4)
    for lb in lbs:
        lmask=(1<<lb)-1
        e00=((b & ~lmask)<<2)|(b&lmask)
        rows[lb]=np.concatenate([(e00|(k<<lb)).reshape(-1,32) for k in range(4)])
    if last is not None:
        lb=last; lmask=(1<<lb)-1; bb=np.arange(TS//2)
        e0=((bb & ~lmask)<<1)|(bb&lmask)
        rows['last']=np.concatenate([e0.reshape(-1,32),(e0|(1<<lb)).reshape(-1,32)])
    rows['linear']=np.arange(TS).reshape(-1,32)
    return rows
def score(rows,padf):
    per={}
    for name,E in rows.items():
        a=(9*E+padf(E))%32
        s=np.sort(a,axis=1); m=np.ones(len(s),dtype=int); run=np.ones(len(s),dtype=int)
        for c in range(1,32):
            same=s[:,c]==s[:,c-1]; run=np.where(same,run+1,1); m=np.maximum(m,run)
        per[name]=round(float((m-1).mean()),2)
    return per
for label,rows in [('first pass 2^11 tile (lb 0,2,4,6,8 + radix-2 at 10)', rows_for((0,2,4,6,8),10)),
                   ('second pass 2^10 stages x 2 columns (lb 1,3,5,7,9)', rows_for((1,3,5,7,9))),
                   ('S22: 11+11, second pass lb 0..: (0,2,4,6,8)+last 10', rows_for((0,2,4,6,8),10)),
                   ('small tile 1024 (lb 0,2,4,6,8)', rows_for((0,2,4,6,8),None,1024,256)),
                   ('small tile 1024 with column bit (lb 1,3,5,7 + last 9)', rows_for((1,3,5,7),9,1024,256))]:
    print(label)
    for name,f in [('e>>4 (rounds 1-5)',lambda E:E>>4),('e>>5 (now)',lambda E:E>>5),('none',lambda E:0*E),('e>>6',lambda E:E>>6),('(e>>5)+(e>>6)',lambda E:(E>>5)+(E>>6)),('3*(e>>5)',lambda E:3*(E>>5))]:
        p=score(rows,f); print('   %-16s mean %.3f  %s'%(name, sum(p.values())/len(p), p))
