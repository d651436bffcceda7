import re,sys
def analyze(f, kidx=0):
    lines=open(f).read().split('\n')
    starts=[i for i,l in enumerate(lines) if l.startswith('_ZN2ba7k_align') and '@' in l]
    s=starts[kidx]; e=next(i for i in range(s,len(lines)) if lines[i].startswith('.Lfunc_end'))
    blocks=[]; cur=('entry',[],'')
    for l in lines[s+1:e]:
        t=l.strip()
        m=re.match(r'^(\.LBB\d+_\d+):(.*)',t)
        if m:
            blocks.append(cur); cur=(m.group(1),[],m.group(2))
        elif t and not t.startswith(';') and not t.startswith('.'):
            cur[1].append(t)
    blocks.append(cur)
    tot=0
    for n,ins,c in blocks:
        ops=[x.split()[0] for x in ins]
        scr=sum(1 for x in ops if x.startswith('scratch_'))
        dpp=sum(1 for x in ins if 'v_max_i32_dpp' in x)
        v=sum(1 for x in ops if x.startswith('v_'))
        depth=re.search(r'Depth=(\d)',c)
        if scr: print(f"  {n} depth={depth.group(1) if depth else '-'} V={v} dpp={dpp} scratch={scr}")
        tot+=scr
    print(f, 'total scratch', tot)
analyze(sys.argv[1], int(sys.argv[2]) if len(sys.argv)>2 else 0)
