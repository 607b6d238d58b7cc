import sys,re,subprocess
src=sys.argv[1]
out=subprocess.run(["/opt/rocm/bin/hipcc","-O3","-std=c++17","-fPIC","--offload-arch=gfx950","-ffp-contract=off","-c",src,"-o","/tmp/kru.o","-Rpass-analysis=kernel-resource-usage"]+sys.argv[2:],capture_output=True,text=True).stderr
cur=None
for line in out.splitlines():
    m=re.search(r"remark: (.*) \[-Rpass",line)
    if not m: 
        if 'error' in line: print(line)
        continue
    t=m.group(1).strip()
    if t.startswith("Function Name:"):
        name=t.split(":",1)[1].strip()
        name=subprocess.run(["c++filt",name],capture_output=True,text=True).stdout.strip()
        cur=name[:90]; vals={}
    else:
        k,v=t.split(":",1); vals[k.strip()]=v.strip()
        if k.strip().startswith("LDS Size"):
            print(f"{cur:90s} VGPR {vals.get('VGPRs')} AGPR {vals.get('AGPRs')} spillV {vals.get('VGPRs Spill')} spillS {vals.get('SGPRs Spill')} scratch {vals.get('ScratchSize [bytes/lane]')} occ {vals.get('Occupancy [waves/SIMD]')} sgpr {vals.get('TotalSGPRs')}")
