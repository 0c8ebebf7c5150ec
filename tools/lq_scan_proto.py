import numpy as np
rng = np.random.default_rng(0)
nx, nu, H = 2, 1, 20
def sym(M): return 0.5*(M+M.T)
def make():
    A = rng.normal(size=(H,nx,nx))*0.5 + np.eye(nx); B = rng.normal(size=(H,nx,nu)); c = rng.normal(size=(H,nx))*0.1
    W = np.array([sym(rng.normal(size=(nx+nu,nx+nu)))*0.3 for _ in range(H)])
    Qs = 2*np.eye(nx); Rs = 0.2*np.eye(nu); QTs = 2*np.eye(nx)
    gr = rng.normal(size=(H*(nx+nu))); bh = rng.uniform(0,0.5,size=H*(nx+nu))
    return A,B,c,W,Qs,Rs,QTs,gr,bh
def seq(A,B,c,W,Qs,Rs,QTs,gr,bh,reg):
    uo=H*nx
    P = QTs + np.diag(bh[(H-1)*nx:H*nx]); p = gr[(H-1)*nx:H*nx].copy()
    Ps=[None]*H; ps=[None]*H; Ks=[None]*H; ks=[None]*H
    for t in range(H-1,-1,-1):
        Ps[t]=P; ps[t]=p
        PA=P@A[t]; PB=P@B[t]; Pc=P@c[t]+p
        Quu = Rs + W[t][nx:,nx:] + np.diag(bh[uo+t*nu:uo+(t+1)*nu]) + reg*np.eye(nu) + B[t].T@PB
        qu = gr[uo+t*nu:uo+(t+1)*nu] + B[t].T@Pc
        Qux = W[t][nx:,:nx] + B[t].T@PA
        if np.linalg.eigvalsh(Quu).min() <= 1e-12: return None
        K = -np.linalg.solve(Quu,Qux); kv = -np.linalg.solve(Quu,qu)
        Ks[t]=K; ks[t]=kv
        if t>0:
            Pn = Qs + W[t][:nx,:nx] + np.diag(bh[(t-1)*nx:t*nx]) + A[t].T@PA + Qux.T@K
            p = gr[(t-1)*nx:t*nx] + A[t].T@Pc + Qux.T@kv
            P = sym(Pn)
    dx=np.zeros(nx); dz=np.zeros(H*(nx+nu)); lam=np.zeros(H*nx)
    for t in range(H):
        du = ks[t] + (Ks[t]@dx if t>0 else 0)
        dxn = c[t] + (A[t]@dx if t>0 else 0) + B[t]@du
        lam[t*nx:(t+1)*nx] = ps[t] + Ps[t]@dxn
        dz[t*nx:(t+1)*nx]=dxn; dz[uo+t*nu:uo+(t+1)*nu]=du; dx=dxn
    return dz,lam
def combine(i,j):
    Ai,bi,Ci,ei,Ji = i; Aj,bj,Cj,ej,Jj = j
    M = np.linalg.inv(np.eye(nx)+Ci@Jj)
    AM = Aj@M
    return (AM@Ai, AM@(bi+Ci@ej)+bj, AM@Ci@Aj.T+Cj, Ai.T@M.T@(ej-Jj@bi)+ei, Ai.T@M.T@Jj@Ai+Ji)
def scan(A,B,c,W,Qs,Rs,QTs,gr,bh,reg,sigma=1.0):
    uo=H*nx
    E=[]
    for t in range(H):
        Bt=B[t]; At=A[t]
        U = Rs + W[t][nx:,nx:] + np.diag(bh[uo+t*nu:uo+(t+1)*nu]) + reg*np.eye(nu) + sigma*Bt.T@Bt
        wux = W[t][nx:,:nx] + sigma*Bt.T@At
        qu = gr[uo+t*nu:uo+(t+1)*nu] + sigma*Bt.T@c[t]
        if t>0:
            Wxx = Qs + W[t][:nx,:nx] + np.diag(bh[(t-1)*nx:t*nx]) + sigma*At.T@At - sigma*np.eye(nx)
            qx = gr[(t-1)*nx:t*nx] + sigma*At.T@c[t]
        else:
            Wxx = np.zeros((nx,nx)); qx=np.zeros(nx)
        if np.linalg.eigvalsh(U).min()<=1e-12: return None
        iU = np.linalg.inv(U)
        E.append((At - Bt@iU@wux, c[t]-Bt@iU@qu, Bt@iU@Bt.T, -(qx - wux.T@iU@qu), Wxx - wux.T@iU@wux))
    E.append((np.zeros((nx,nx)), np.zeros(nx), np.zeros((nx,nx)), -gr[(H-1)*nx:H*nx], QTs+np.diag(bh[(H-1)*nx:H*nx]) - sigma*np.eye(nx)))
    # Hillis-Steele suffix scan
    d=1
    while d < H+1:
        newE=list(E)
        for e in range(H+1):
            if e+d <= H: newE[e]=combine(E[e],E[e+d])
        E=newE; d*=2
    Ks=[];ks=[];Ps=[];ps=[];F=[];f=[]
    for t in range(H):
        P = E[t+1][4] + sigma*np.eye(nx); p = -E[t+1][3]
        PA=P@A[t]; PB=P@B[t]; Pc=P@c[t]+p
        Quu = Rs + W[t][nx:,nx:] + np.diag(bh[uo+t*nu:uo+(t+1)*nu]) + reg*np.eye(nu) + B[t].T@PB
        if np.linalg.eigvalsh(Quu).min() <= 1e-12: return None
        qu = gr[uo+t*nu:uo+(t+1)*nu] + B[t].T@Pc
        Qux = W[t][nx:,:nx] + B[t].T@PA
        K=-np.linalg.solve(Quu,Qux); kv=-np.linalg.solve(Quu,qu)
        Ks.append(K);ks.append(kv);Ps.append(P);ps.append(p)
        F.append(A[t]+B[t]@K); f.append(c[t]+B[t]@kv)
    F[0]=np.zeros((nx,nx))   # dx_0 = 0
    d=1
    while d<H:
        nF=list(F); nf=list(f)
        for e in range(H):
            if e-d>=0: nF[e]=F[e]@F[e-d]; nf[e]=F[e]@f[e-d]+f[e]
        F=nF;f=nf;d*=2
    dz=np.zeros(H*(nx+nu)); lam=np.zeros(H*nx)
    for t in range(H):
        dx = f[t-1] if t>0 else np.zeros(nx)
        du = ks[t]+Ks[t]@dx
        dxn = f[t]
        dz[t*nx:(t+1)*nx]=dxn; dz[uo+t*nu:uo+(t+1)*nu]=du; lam[t*nx:(t+1)*nx]=ps[t]+Ps[t]@dxn
    return dz,lam
ok=0; worst=0
for trial in range(200):
    p = make()
    r1 = seq(*p, reg=1e-9); r2 = scan(*p, reg=1e-9)
    if r1 is None or r2 is None:
        print(trial, "seq", r1 is None, "scan", r2 is None); continue
    e = max(np.abs(r1[0]-r2[0]).max()/max(1,np.abs(r1[0]).max()), np.abs(r1[1]-r2[1]).max()/max(1,np.abs(r1[1]).max()))
    worst=max(worst,e); ok+=1
print(ok, worst)

print("---- failure categories")
for scale in (0.3, 0.6, 1.0):
    for sigma in (0.0, 1.0, 3.0):
        rng = np.random.default_rng(1)
        def make2():
            A = rng.normal(size=(H,nx,nx))*0.5 + np.eye(nx); B = rng.normal(size=(H,nx,nu))*0.5; c = rng.normal(size=(H,nx))*0.1
            W = np.array([sym(rng.normal(size=(nx+nu,nx+nu)))*scale for _ in range(H)])
            return A,B,c,W,2*np.eye(nx),0.2*np.eye(nu),2*np.eye(nx),rng.normal(size=(H*(nx+nu))),rng.uniform(0,0.5,size=H*(nx+nu))
        cnt = {"both":0,"seq_only":0,"scan_only":0,"none":0}; worst=0
        for trial in range(300):
            p = make2()
            for reg in (1e-9, 1e-3, 1e-2, 1e-1, 1.0, 10.0):
                r1 = seq(*p, reg=reg); 
                try: r2 = scan(*p, reg=reg, sigma=sigma)
                except np.linalg.LinAlgError: r2=None
                if r1 is not None and r2 is not None:
                    e = max(np.abs(r1[0]-r2[0]).max()/max(1,np.abs(r1[0]).max()), np.abs(r1[1]-r2[1]).max()/max(1,np.abs(r1[1]).max()))
                    worst=max(worst,e)
                k = "both" if (r1 is not None and r2 is not None) else "seq_only" if r1 is not None else "scan_only" if r2 is not None else "none"
                cnt[k]+=1
        print(scale, sigma, cnt, "worst rel err %.2e"%worst)
