import sys,re,collections
agg=collections.defaultdict(list)
for l in sys.stdin:
    m=re.search(r"\[wreg stamps\] (M=\d+ N=\d+ Cin=\d+ sk=\d+ cps=\d+ wgs=\d+) (3-tile wave|staging wave)\s+\| avg ticks: setup\+table (\d+), first halo (\d+) \(landing (\d+), pass (\d+), next DMA \+ barrier (\d+)\), loop (\d+) \((\d+) per tap\), k-half sum (\d+), stores (\d+)",l)
    if m: agg[(m.group(1),m.group(2))].append([int(x) for x in m.groups()[2:]])
for k,v in agg.items():
    n=len(v); a=[sum(x[i] for x in v)/n for i in range(9)]
    print(k[0],k[1][:5],n,'setup %.0f halo %.0f (land %.0f pass %.0f next %.0f) loop %.0f (%.0f/tap) khalf %.0f stores %.0f total %.0f'%(a[0],a[1],a[2],a[3],a[4],a[5],a[6],a[7],a[8],a[0]+a[1]+a[5]+a[7]+a[8]))
