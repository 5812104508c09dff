"""Debug dump for tests/test_gpu_round5.py: backward_qt_bf16 / mxfp4_transpose_mxfp8 on operands with e8m0 bytes 0 / 255 (run on the GPU box)."""
import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np, torch, oracle
ns={}; exec(compile(open('tests/test_gpu_round5.py').read(),'t5','exec'),ns)
import qutlass_amd as q
_np=ns['_np']; DEV='cuda:0'
B,N,M=1,96,64
rng=np.random.default_rng(N+M)
codes=rng.integers(0,256,size=(B,N,M//2),dtype=np.uint8); scales=rng.integers(118,134,size=(B,N,M//32),dtype=np.uint8)
scales[:,5::64,0]=0; scales[:,40::64,1]=255; scales[:,7::96,-1]=0
h=ns['_hadamard'](32); alpha=torch.tensor([0.61],device=DEV)
e2m1,e8m0=q.backward_qt_bf16(torch.from_numpy(codes).to(DEV),torch.from_numpy(scales).to(DEV).view(torch.float8_e8m0fnu),h,alpha)
rq,rs=oracle.backward_qt_bf16(codes,scales,_np(h),0.61,acc_model=1)
gs=_np(e8m0).reshape(rs.shape)
bad=np.argwhere(gs!=rs)
print('QT scale shape',rs.shape,'bad',len(bad))
for b in bad[:12]:
    b=tuple(b); print('  at',b,'got',gs[b],'ref',rs[b])   # output (B, M, N/32): m index b[1], group b[2] -> input rows 32*b[2].., column m
    m,g=b[1],b[2]
    print('     input scale bytes of rows',32*g,'..:',scales[0,32*g:32*g+32,m//32].tolist())
m,n=256,512
rng=np.random.default_rng(m+n)
codes=rng.integers(0,256,size=(m,n//2),dtype=np.uint8); scales=rng.integers(117,137,size=(m,n//32),dtype=np.uint8)
scales[3::32,0]=0; scales[17::64,2]=0; scales[9::64,1]=255
y,sf=q.mxfp4_transpose_mxfp8(torch.from_numpy(codes).to(DEV),torch.from_numpy(scales).to(DEV).view(torch.float8_e8m0fnu))
ry,rs=oracle.mxfp4_transpose_mxfp8(codes,scales)
gy=_np(y).reshape(n,-1); ry=np.asarray(ry).reshape(n,-1); gsf=_np(sf).reshape(n,-1); rs=np.asarray(rs).reshape(n,-1)
print('TR shapes',gy.shape,ry.shape,gsf.shape,rs.shape)
bad=np.argwhere(gsf!=rs); print('TR scale bad',len(bad))
for b in bad[:8]:
    b=tuple(b); col,g=b; print('  col',col,'grp',g,'got',gsf[b],'ref',rs[b],'input scales',scales[32*g:32*g+32,col//32].tolist())
bad=np.argwhere(gy!=ry); print('TR y bad',len(bad))
for b in bad[:12]:
    b=tuple(b); col,r=b; print('  col',col,'row',r,'got %02x ref %02x'%(gy[b],ry[b]),'in scale',scales[r,col//32],'code',(codes[r,col//2]>>(4*(col&1)))&15, 'out scale got/ref',gsf[col,r//32],rs[col,r//32])
