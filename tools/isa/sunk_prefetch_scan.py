import re,sys
lines=open(sys.argv[1]).read().split('\n')
name=sys.argv[2]
start=[i for i,l in enumerate(lines) if name in l and l.endswith('>:')][0]
ends=[i for i in range(start+1,len(lines)) if lines[i].endswith('>:')]
end=ends[0] if ends else len(lines)
body=[l.strip().split('//')[0].rstrip() for l in lines[start+1:end] if l.strip()]
loads=[]  # indices of vmem loads in order
hits=[]
for i,x in enumerate(body):
    if x.startswith(('buffer_load','global_load','scratch_load','flat_load')): loads.append(i); continue
    if x.startswith(('s_cbranch','s_branch','s_barrier')): loads=[]; continue   # new region: unknown
    m=re.match(r's_waitcnt vmcnt\((\d+)\)',x)
    if m:
        n=int(m.group(1))
        if len(loads)>n:
            tgt=loads[-1-n]
            if i-tgt<=16: hits.append((i,i-tgt,n))
        loads=loads[-n:] if n>0 else []
# cluster and show those with MFMAs nearby
cl=[]
for h in hits:
    if cl and h[0]-cl[-1][-1][0]<80: cl[-1].append(h)
    else: cl.append([h])
print(name,len(body),'waits on a load issued <=16 instrs earlier:',len(hits))
for c in cl:
    a,b=c[0][0],c[-1][0]
    mf=sum(1 for x in body[max(0,a-100):b+100] if x.startswith('v_mfma'))
    if len(c)>=3 and mf>20: print('  ',a,b,len(c),'mfma nearby',mf)
