import json,sys
d=json.load(open('bench_full.json'))
k=d['fp64_valu']['kernel_ms']; a=d['fp64_valu'].get('kernel_ms_alone')
print(d['config']['tag'], 'ms/step %.3f'%d['ms_per_step'], 'K1 %.2f K2 %.2f K3 %.2f K3b %.2f'%(k['k_singlet'],k['k_doublet'],k['k_reduce'],k['k_certify']), 'alone', {x:round(y,2) for x,y in (a or {}).items()}, d['kernels_launched'])
