# per-kernel register / scratch / occupancy / instruction counts from a hipcc -S --cuda-device-only listing
import re, sys
name = None; n = 0; rows = []
for line in open(sys.argv[1]):
    m = re.match(r'^(_Z\w+):', line)
    if m: name = m.group(1); n = 0; info = {}; continue
    if name is None: continue
    if re.match(r'^\t[a-z]', line) and not line.startswith('\t.'): n += 1
    for key in ('NumVgprs', 'ScratchSize', 'Occupancy', 'NumSgprs', 'LDSByteSize'):
        m = re.match(r'^; %s: (\d+)' % key, line)
        if m: info[key] = int(m.group(1))
    if line.startswith('; Occupancy'):
        rows.append((name, n, info)); name = None
for name, n, info in rows:
    if len(sys.argv) > 2 and sys.argv[2] not in name: continue
    print('%-70s instrs %5d vgpr %3d sgpr %3d scratch %4d lds %6d occ %d' % (name[:70], n, info.get('NumVgprs', -1), info.get('NumSgprs', -1), info.get('ScratchSize', -1), info.get('LDSByteSize', -1), info.get('Occupancy', -1)))
