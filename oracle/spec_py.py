"""spec_py.py — executable specification (test infrastructure, NOT product code).

Pure-Python/numpy restatement of the spoa 4.0.8 subset + racon's
Window::generate_consensus driver (reference src/window.cpp:65-149) and of the
edlib 1.2.7 path-selection rules used by reference src/overlap.cpp:205-224.
Taken from SURVEY.md Appendix D (the survey's emulation that reproduced all ten
CPU goldens of reference test/racon_test.cpp:86-295).  Used only by tests/ as an
independent second implementation to cross-check oracle/poa_oracle.cpp on
small cases.  Slow by design.
"""
import sys, numpy as np
# ---------- libstdc++ std::sort emulation (introsort, threshold 16) ----------
def std_sort(a, lo, hi, less):
    def lg(n): return n.bit_length()-1
    def move_median_to_first(r,x,y,z):
        if less(a[x],a[y]):
            if less(a[y],a[z]): a[r],a[y]=a[y],a[r]
            elif less(a[x],a[z]): a[r],a[z]=a[z],a[r]
            else: a[r],a[x]=a[x],a[r]
        elif less(a[x],a[z]): a[r],a[x]=a[x],a[r]
        elif less(a[y],a[z]): a[r],a[z]=a[z],a[r]
        else: a[r],a[y]=a[y],a[r]
    def unguarded_partition(f,l,p):
        while True:
            while less(a[f],a[p]): f+=1
            l-=1
            while less(a[p],a[l]): l-=1
            if not f<l: return f
            a[f],a[l]=a[l],a[f]; f+=1
    def heap_sort(f,l):
        sub=a[f:l]
        # partial_sort(first,last,last) == heap sort; rare (depth limit). emulate roughly with stable sort (flag it)
        print('WARNING heap sort path hit',file=sys.stderr)
        import functools
        sub.sort(key=functools.cmp_to_key(lambda x,y:-1 if less(x,y) else (1 if less(y,x) else 0))); a[f:l]=sub
    def introsort_loop(f,l,depth):
        while l-f>16:
            if depth==0: heap_sort(f,l); return
            depth-=1
            mid=f+(l-f)//2
            move_median_to_first(f,f+1,mid,l-1)
            cut=unguarded_partition(f+1,l,f)
            introsort_loop(cut,l,depth)
            l=cut
    def unguarded_linear_insert(i):
        v=a[i]; n=i-1
        while less(v,a[n]): a[i]=a[n]; i=n; n-=1
        a[i]=v
    def insertion_sort(f,l):
        if f==l: return
        for i in range(f+1,l):
            if less(a[i],a[f]):
                v=a[i]; a[f+1:i+1]=a[f:i]; a[f]=v
            else: unguarded_linear_insert(i)
    if lo==hi: return
    introsort_loop(lo,hi,lg(hi-lo)*2)
    if hi-lo>16:
        insertion_sort(lo,lo+16)
        for i in range(lo+16,hi): unguarded_linear_insert(i)
    else: insertion_sort(lo,hi)

# ---------- spoa::Graph (recalled) ----------
class Graph:
    def __init__(s):
        s.coder={}; s.decoder=[]; s.nseq=0
        s.code=[]; s.inn=[]; s.out=[]; s.aln=[]          # per node
        s.et=[]; s.eh=[]; s.ew=[]; s.el=[]               # per edge
        s.rank=[]; 
    def add_node(s,c):
        s.code.append(c); s.inn.append([]); s.out.append([]); s.aln.append([]); return len(s.code)-1
    def add_edge(s,t,h,w):
        for e in s.out[t]:
            if s.eh[e]==h: s.el[e].append(s.nseq); s.ew[e]+=w; return
        e=len(s.et); s.et.append(t); s.eh.append(h); s.ew.append(w); s.el.append([s.nseq])
        s.out[t].append(e); s.inn[h].append(e)
    def add_sequence(s,seq,w,b,e):
        if b==e: return None
        prev=None; first=None
        for i in range(b,e):
            c=s.add_node(s.coder[seq[i]])
            if first is None: first=c
            if prev is not None: s.add_edge(prev,c,w[i-1]+w[i])
            prev=c
        return first
    def add_alignment(s,al,seq,w):
        if len(seq)==0: return
        for ch in seq:
            if ch not in s.coder: s.coder[ch]=len(s.decoder); s.decoder.append(ch)
        if not al:
            s.add_sequence(seq,w,0,len(seq)); s.nseq+=1; s.toposort(); return
        valid=[b for (a,b) in al if b!=-1]
        assert valid
        begin=s.add_sequence(seq,w,0,valid[0])
        prev=len(s.code)-1 if begin is not None else None
        last=s.add_sequence(seq,w,valid[-1]+1,len(seq))
        for (a,b) in al:
            if b==-1: continue
            c=s.coder[seq[b]]; curr=None
            if a==-1: curr=s.add_node(c)
            else:
                if s.code[a]==c: curr=a
                else:
                    for k in s.aln[a]:
                        if s.code[k]==c: curr=k; break
                    if curr is None:
                        curr=s.add_node(c)
                        for k in s.aln[a]:
                            s.aln[k].append(curr); s.aln[curr].append(k)
                        s.aln[a].append(curr); s.aln[curr].append(a)
            if begin is None: begin=curr
            if prev is not None: s.add_edge(prev,curr,w[b-1]+w[b])
            prev=curr
        if last is not None: s.add_edge(prev,last,w[valid[-1]]+w[valid[-1]+1])
        s.nseq+=1; s.toposort()
    def toposort(s):
        n=len(s.code); marks=[0]*n; ign=[False]*n; rank=[]
        for start in range(n):
            if marks[start]!=0: continue
            st=[start]
            while st:
                c=st[-1]; ok=True
                if marks[c]!=2:
                    for e in s.inn[c]:
                        t=s.et[e]
                        if marks[t]!=2: st.append(t); ok=False
                    if not ign[c]:
                        for k in s.aln[c]:
                            if marks[k]!=2: st.append(k); ign[k]=True; ok=False
                    if ok:
                        marks[c]=2
                        if not ign[c]:
                            rank.append(c); rank.extend(s.aln[c])
                    else: marks[c]=1
                if ok: st.pop()
        assert len(rank)==n, (len(rank),n)
        s.rank=rank
    def subgraph(s,begin,end):
        n=len(s.code); inc=[False]*n; st=[end]
        while st:
            c=st.pop()
            if not inc[c] and c>=begin:
                for e in s.inn[c]: st.append(s.et[e])
                for k in s.aln[c]: st.append(k)
                inc[c]=True
        g=Graph(); g.coder=s.coder; g.decoder=s.decoder
        s2g=[]; g2s=[None]*n
        for i in range(n):
            if inc[i]: g2s[i]=g.add_node(s.code[i]); s2g.append(i)
        for i in range(n):
            if not inc[i]: continue
            j=g2s[i]
            for e in s.inn[i]:
                t=g2s[s.et[e]]
                if t is not None: g.add_edge(t,j,s.ew[e])
            for k in s.aln[i]:
                if g2s[k] is not None: g.aln[j].append(g2s[k])
        g.toposort(); return g,s2g
    def coverage(s,v):
        lab=set()
        for e in s.inn[v]: lab.update(s.el[e])
        for e in s.out[v]: lab.update(s.el[e])
        return len(lab)
    def consensus(s):
        n=len(s.code); pred=[None]*n; sc=[-1]*n; mx=None
        def relax(it,skip):
            for e in s.inn[it]:
                t=s.et[e]
                if skip and sc[t]==-1: continue
                w=s.ew[e]
                if sc[it]<w or (sc[it]==w and sc[pred[it]]<=sc[t]): sc[it]=w; pred[it]=t
            if pred[it] is not None: sc[it]+=sc[pred[it]]
        for it in s.rank:
            relax(it,False)
            if mx is None or sc[mx]<sc[it]: mx=it
        if s.out[mx]:
            n2r=[0]*n
            for i,v in enumerate(s.rank): n2r[v]=i
            while s.out[mx]:
                start=mx
                for e in s.out[start]:
                    for f in s.inn[s.eh[e]]:
                        if s.et[f]!=start: sc[s.et[f]]=-1
                m2=None
                for i in range(n2r[start]+1,n):
                    it=s.rank[i]; sc[it]=-1; pred[it]=None
                    relax(it,True)
                    if m2 is None or sc[m2]<sc[it]: m2=it
                mx=m2
        cons=[]
        while pred[mx] is not None: cons.append(mx); mx=pred[mx]
        cons.append(mx); cons.reverse()
        cov=[s.coverage(v)+sum(s.coverage(k) for k in s.aln[v]) for v in cons]
        return ''.join(s.decoder[s.code[v]] for v in cons),cov

# ---------- spoa SISD NW linear (recalled) ----------
def align(seq,g,m,n_,gap):
    V=len(g.code); L=len(seq)
    if V==0 or L==0: return []
    sa=np.frombuffer(seq.encode(),dtype=np.uint8)
    prof=[np.concatenate(([0],np.where(sa==ord(ch),m,n_))).astype(np.int32) for ch in g.decoder]
    n2r=[0]*V
    for i,v in enumerate(g.rank): n2r[v]=i
    H=np.empty((V+1,L+1),dtype=np.int32)
    jg=np.arange(L+1,dtype=np.int32)*gap
    H[0]=jg
    preds=[]
    best=None;bi=0
    for r,v in enumerate(g.rank):
        i=r+1
        ps=[n2r[g.et[e]]+1 for e in g.inn[v]] or [0]
        preds.append(ps)
        P=prof[g.code[v]]
        h0=(max(H[p,0] for p in ps) if g.inn[v] else 0)+gap
        Hp=H[ps[0]]
        row=np.maximum(Hp[:-1]+P[1:],Hp[1:]+gap)
        for p in ps[1:]:
            Hp=H[p]
            row=np.maximum(row,np.maximum(Hp[:-1]+P[1:],Hp[1:]+gap))
        full=np.empty(L+1,dtype=np.int32); full[0]=h0; full[1:]=row
        full=np.maximum.accumulate(full-jg)+jg
        H[i]=full
        if not g.out[v]:
            if best is None or best<full[L]: best=int(full[L]); bi=i
    i=bi;j=L; al=[]
    while not (i==0 and j==0):
        hij=H[i,j]; found=False
        if i!=0 and j!=0:
            v=g.rank[i-1]; mc=prof[g.code[v]][j]
            for p in preds[i-1]:
                if hij==H[p,j-1]+mc: pi,pj=p,j-1; found=True; break
        if not found and i!=0:
            for p in preds[i-1]:
                if hij==H[p,j]+gap: pi,pj=p,j; found=True; break
        if not found and hij==H[i,j-1]+gap: pi,pj=i,j-1; found=True
        assert found
        al.append((-1 if i==pi else g.rank[i-1], -1 if j==pj else j-1))
        i,j=pi,pj
    al.reverse(); return al

# ---------- racon Window::generate_consensus ----------
def window_consensus(win,m,n_,gap,tgs=True,trim=True):
    seqs=win['seqs']
    if len(seqs)<3: return seqs[0][0],False
    g=Graph()
    def wts(s,q): return [ord(c)-33 for c in q] if q is not None else [1]*len(s)
    g.add_alignment([],seqs[0][0],wts(*seqs[0][:2]))
    rank=list(range(len(seqs)))
    std_sort(rank,1,len(rank),lambda a,b: seqs[a][2]<seqs[b][2])
    Lb=len(seqs[0][0]); off=int(0.01*Lb)
    for i in rank[1:]:
        s,q,b,e=seqs[i]
        if b<off and e>Lb-off: al=align(s,g,m,n_,gap)
        else:
            sg,mp=g.subgraph(b,e); al=align(s,sg,m,n_,gap)
            al=[(mp[a] if a!=-1 else -1,c) for (a,c) in al]
        g.add_alignment(al,s,wts(s,q))
    cons,cov=g.consensus()
    if tgs and trim:
        avg=(len(seqs)-1)//2; b=0; e=len(cons)-1
        while b<len(cons) and cov[b]<avg: b+=1
        while e>=0 and cov[e]<avg: e-=1
        if b>=e: print('chimeric warn',file=sys.stderr)
        else: cons=cons[b:e+1]
    return cons,True


# ---------- edlib path selection ----------
WORD=64
def col_scores(q, t):
    """last column of NW edit-distance matrix of q (rows) vs t (cols): D[0..m][len(t)]"""
    m=len(q); idx=np.arange(m+1,dtype=np.int32)
    prev=idx.copy()
    for j in range(1,len(t)+1):
        neq=(q!=t[j-1]).astype(np.int32)
        tmp=np.empty(m+1,dtype=np.int32); tmp[0]=j
        np.minimum(prev[1:]+1, prev[:-1]+neq, out=tmp[1:])
        prev=np.minimum.accumulate(tmp-idx)+idx
    return prev
def full_matrix(q,t):
    m=len(q); n=len(t); idx=np.arange(m+1,dtype=np.int32)
    D=np.empty((n+1,m+1),dtype=np.int32); D[0]=idx
    for j in range(1,n+1):
        neq=(q!=t[j-1]).astype(np.int32)
        tmp=np.empty(m+1,dtype=np.int32); tmp[0]=j
        np.minimum(D[j-1,1:]+1, D[j-1,:-1]+neq, out=tmp[1:])
        D[j]=np.minimum.accumulate(tmp-idx)+idx
    return D   # D[j][i]
def traceback(q,t):
    m=len(q); n=len(t); D=full_matrix(q,t); i=m; j=n; ops=[]
    while i>0 or j>0:
        cur=D[j,i]
        if i>0 and D[j,i-1]+1==cur: ops.append('I'); i-=1          # up: consumes query
        elif j>0 and D[j-1,i]+1==cur: ops.append('D'); j-=1        # left: consumes target
        else: ops.append('M'); i-=1; j-=1
    ops.reverse(); return ops, int(D[n,m])
def obtain(q,t,best):
    m=len(q); n=len(t)
    if m==0 or n==0: return ['D']*n if m==0 else ['I']*m
    nb=(m+WORD-1)//WORD
    if (2*8+4)*nb*n + 2*4*n < 1024*1024:
        ops,sc=traceback(q,t); assert sc==best,(sc,best); return ops
    lw=n//2; rw=n-lw
    left=col_scores(q,t[:lw])                 # left[i] = ED(q[:i], t[:lw])
    right=col_scores(q[::-1],t[lw:][::-1])    # right[k] = ED(q[m-k:], t[lw:])
    # candidate: queryIdx (0-based row in left column) -> h=queryIdx+1 rows on the left; right gets q[h:]
    h=None
    for qi in range(0,m-1):                   # queryIdx from 0 .. m-2  (left cell row qi, right cell row qi+1 exists)
        if left[qi+1]+right[m-(qi+1)]==best: h=qi+1; break
    if h is None and lw+right[m]==best: h=0
    if h is None and left[m]+rw==best: h=m
    assert h is not None
    ls=int(left[h]) if h>0 else lw; rs=int(right[m-h]) if h<m else rw
    return obtain(q[:h],t[:lw],ls)+obtain(q[h:],t[lw:],rs)
# usage: best = col_scores(q,t)[-1]; ops = obtain(q,t,best); CIGAR = run-length encode ops
