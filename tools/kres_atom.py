"""Register / scratch / occupancy table of the atom kernel's instantiations from a -Rpass-analysis=kernel-resource-usage log."""
import re, sys
t = open(sys.argv[1]).read()
for m in re.finditer(r"Function Name: (\S+).*?VGPRs: (\d+).*?AGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?Occupancy \[waves/SIMD\]: (\d+).*?LDS Size", t, re.S):
    mm = re.search(r"k_atom_fwdILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E", m.group(1))
    if not mm:
        continue
    print("C=%-3s NTP=%-3s mode=%s NP=%s : vgpr %3s scratch %3s occ %s" % (mm.group(1), mm.group(2), mm.group(4), mm.group(5), m.group(2), m.group(4), m.group(5)))
