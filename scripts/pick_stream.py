import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stream']
print("py garble %.1f us eval steady %.1f | native big %s"%(s['steady_ms_per_step']*1e3, s['eval_steady_ms_per_step']*1e3, {k:round(v,1) for k,v in s['native_host']['big130'].items() if 'us_per' in k}))
