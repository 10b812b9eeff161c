import sys,re,collections
acc=collections.defaultdict(list)
for l in sys.stdin:
    if 'host wall ms' in l:
        for k,v in re.findall(r'([a-zA-Z/+ ]+?) ([0-9.]+)(?: \||$)', l.split('host wall ms:')[1]): acc['wall '+k.strip()].append(float(v))
    if 'pair selection ms' in l:
        for k,v in re.findall(r'(pass 1|pass 2|order|pass 3\+4) ([0-9.]+)', l): acc['sel '+k].append(float(v))
    if 'GPU stage lock' in l: print(l.strip()[:300])
for k,v in acc.items():
    v=v[len(v)//3:]
    print('%-45s n=%d mean %.2f'%(k,len(v),sum(v)/len(v)))
