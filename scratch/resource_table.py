# Kernel resource table from `hipcc -Rpass-analysis=kernel-resource-usage` remarks (one text file per translation unit).
#   usage: resource_table.py <dir with *.txt> > profiles/rNN_resource_usage.txt
import glob, os, re, subprocess, sys
d = sys.argv[1]
rows = []
for f in sorted(glob.glob(os.path.join(d, '*.txt'))):
    cur = None
    for line in open(f):
        m = re.search(r'^(.*?):(\d+):\d+: remark: Function Name: (\S+)', line)
        if m:
            cur = {'file': os.path.basename(m.group(1)), 'line': int(m.group(2)), 'name': m.group(3)}
            rows.append(cur)
            continue
        m = re.search(r'remark:\s+(TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Dynamic Stack|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)', line)
        if m and cur is not None:
            cur[m.group(1)] = m.group(2)
names = subprocess.run(['c++filt'], input='\n'.join(r['name'] for r in rows), capture_output=True, text=True).stdout.splitlines()
print(f"{'kernel':86s} {'file:line':22s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'LDS(static)':>11s} {'scratch':>7s} {'occ':>4s} {'sgpr_spill':>10s} {'vgpr_spill':>10s}")
for r, n in zip(rows, names):
    n = n.replace('lpgp::', '').replace('void ', '')
    n = n[:n.index('(')] if '(' in n else n
    print(f"{n[:86]:86s} {r['file'] + ':' + str(r['line']):22s} {r.get('VGPRs', '?'):>5s} {r.get('AGPRs', '?'):>5s} {r.get('TotalSGPRs', '?'):>5s} "
          f"{r.get('LDS Size [bytes/block]', '?'):>11s} {r.get('ScratchSize [bytes/lane]', '?'):>7s} {r.get('Occupancy [waves/SIMD]', '?'):>4s} "
          f"{r.get('SGPRs Spill', '?'):>10s} {r.get('VGPRs Spill', '?'):>10s}")
