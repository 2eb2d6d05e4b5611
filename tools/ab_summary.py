"""mean / individual values per variant from tools/ab_bench.sh's output"""
import collections
import sys
two, one = collections.defaultdict(list), collections.defaultdict(list)
for line in open(sys.argv[1]):
    p = line.split()
    if len(p) >= 11 and p[2] == "two":
        two[p[0]].append(float(p[5]))
        one[p[0]].append(float(p[10]))
for k in two:
    print(f"{k:8s} two in flight {sum(two[k]) / len(two[k]):7.3f}  ({' '.join(f'{x:.2f}' for x in two[k])})   one at a time {sum(one[k]) / len(one[k]):6.3f}")
