WL=demo_181c3_thick4xyz UNIRES_P2_WTAB=0 python tools/r6_cmp.py build/ab/c1.so 2>&1 | tail -6
echo wtab1
WL=demo_181c3_thick4xyz UNIRES_P2_WTAB=1 python tools/r6_cmp.py build/ab/c1.so 2>&1 | tail -6
