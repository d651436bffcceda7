import re,sys
def analyze(f, kidx):
    lines=open(f).read().split('\n')
    starts=[i for i,l in enumerate(lines) if l.startswith('_ZN2ba7k_align') and '@' in l]
    s=starts[kidx]; e=next(i for i in range(s,len(lines)) if lines[i].startswith('.Lfunc_end'))
    blocks=[]; cur=['entry',[],'']
    for l in lines[s+1:e]:
        t=l.strip()
        m=re.match(r'^(\.LBB\d+_\d+):(.*)',t)
        if m:
            blocks.append(cur); cur=[m.group(1),[],m.group(2)]
        elif t and not t.startswith(';') and not t.startswith('.'):
            cur[1].append(t)
    blocks.append(cur)
    def st(ins):
        ops=[x.split()[0] for x in ins]
        return dict(V=sum(1 for x in ops if x.startswith('v_') and not x.startswith(('v_readlane','v_writelane','v_readfirstlane'))),
                    X=sum(1 for x in ops if x.startswith(('v_readlane','v_writelane','v_readfirstlane'))),
                    scr=sum(1 for x in ops if x.startswith('scratch_')), dpp=sum(1 for x in ins if 'v_max_i32_dpp' in x),
                    S=sum(1 for x in ops if x.startswith('s_') and not x.startswith(('s_nop','s_waitcnt'))))
    # fast path = longest run of >= 8 consecutive dpp==6 blocks (allowing tiny blocks in between) with no loop (unrolled)
    stats=[(b[0],st(b[1]),b[2]) for b in blocks]
    runs=[]; i=0
    while i<len(stats):
        if stats[i][1]['dpp']==6:
            j=i; cols=[]
            while j<len(stats) and (stats[j][1]['dpp']==6 or (stats[j][1]['V']<=6 and stats[j][1]['dpp']==0)):
                if stats[j][1]['dpp']==6: cols.append(j)
                j+=1
            runs.append(cols); i=j
        else: i+=1
    best=max(runs,key=len)
    tot=dict(V=0,X=0,scr=0,S=0)
    for j in best:
        for k in tot: tot[k]+=stats[j][1][k]
    print(f"{f} kernel{kidx}: fast path {len(best)} column blocks {stats[best[0]][0]}..{stats[best[-1]][0]}: {tot}")
    tv=sum(x[1]['V'] for x in stats); ts=sum(x[1]['scr'] for x in stats); tx=sum(x[1]['X'] for x in stats)
    print(f"   whole kernel: V={tv} X={tx} scratch={ts} blocks={len(stats)}")
for k in (0,2): analyze(sys.argv[1], k)
