"""Debug dump for tests/test_gpu_round5.py: the groups of the special-activation inputs where the fused quantizers and the oracle disagree (run on the GPU box)."""
import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np, torch, oracle
ns={}; exec(compile(open('tests/test_gpu_round5.py').read(),'t5','exec'),ns)
import qutlass_amd as q
_np=ns['_np']
def show(tag,x,got_q,got_s,rq,rs,gsz):
    got=got_q.reshape(-1); ref=rq.reshape(-1)
    nb=gsz//2
    bad=[g for g in range(got.size//nb) if not (np.array_equal(got[g*nb:(g+1)*nb],ref[g*nb:(g+1)*nb]) and got_s[g]==rs[g])]
    print(tag,'bad groups',bad[:24])
    for g in bad[:6]:
        print('  group',g,'x bits',' '.join('%04x'%v for v in _np(x).reshape(-1)[g*gsz:(g+1)*gsz]))
        print('   got',' '.join('%02x'%v for v in got[g*nb:(g+1)*nb]),'scale',got_s[g])
        print('   ref',' '.join('%02x'%v for v in ref[g*nb:(g+1)*nb]),'scale',rs[g])
which=sys.argv[1] if len(sys.argv)>1 else 'mx'
if which=='mx':
  for ident in (True, False):
    for method in ("abs_max","quest"):
      x=ns['_special_activations'](8,1024,32,11*32+(method=="quest"))
      h=torch.eye(32,dtype=torch.bfloat16,device='cuda') if ident else ns['_hadamard'](32)
      out=q.fusedQuantizeMx(x,h,method=method)
      rq,rs,rm=oracle.fused_quantize_mx(_np(x),_np(h),oracle.QUEST if method=="quest" else oracle.ABS_MAX,with_mask=False)
      show(f'MX ident={ident} {method}',x,_np(out[0]),_np(out[1]).reshape(-1),rq,rs,32)
else:
  for method in ("abs_max","quest"):
    x=ns['_special_activations'](8,1024,16,7*16+(method=="quest"))
    h=torch.eye(16,dtype=torch.bfloat16,device='cuda')
    gs=torch.tensor([1.5],device='cuda')
    e2m1,e4m3=q.fusedQuantizeNv(x,h,gs,method=method)
    print('shapes',e2m1.shape,e4m3.shape,e4m3.dtype)
    rq,rs=oracle.fused_quantize_nv(_np(x),_np(h),1.5,oracle.QUEST if method=="quest" else oracle.ABS_MAX)
    show(f'NV {method}',x,_np(e2m1),_np(e4m3).reshape(-1),rq,rs,16)
