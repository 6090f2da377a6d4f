#!/usr/bin/env python3
"""Reflow a markdown file to <= WIDTH columns: paragraphs and list items are re-wrapped (continuation lines indented
under the item's text), tables with a row longer than WIDTH become bullet lists ("* **first cell** - second cell; ..."
with the header's column names in front of later cells), code fences and short tables are left alone.
usage: tools/reflow_md.py [--keep-tables] FILE... (in place)"""
import re
import sys
import textwrap

WIDTH = 160
KEEP_TABLES = '--keep-tables' in sys.argv


def wrap(text, first, rest):
    return textwrap.wrap(text, WIDTH, initial_indent=first, subsequent_indent=rest, break_long_words=False,
                         break_on_hyphens=False) or [first.rstrip()]


def cells(row):
    row = row.strip()
    if row.startswith('|'):
        row = row[1:]
    if row.endswith('|'):
        row = row[:-1]
    out, cur, code = [], '', False
    i = 0
    while i < len(row):
        ch = row[i]
        if ch == '`':
            code = not code
        if ch == '\\' and i + 1 < len(row) and row[i + 1] == '|':
            cur += '|'
            i += 2
            continue
        if ch == '|' and not code:
            out.append(cur.strip())
            cur = ''
        else:
            cur += ch
        i += 1
    out.append(cur.strip())
    return out


def table_to_list(rows):
    head = cells(rows[0])
    out = []
    for r in rows[2:]:
        c = cells(r)
        text = '**%s**' % c[0] if c and c[0] else ''
        for k, v in enumerate(c[1:], 1):
            if not v:
                continue
            name = head[k] if k < len(head) else ''
            sep = ' - ' if k == 1 else '; '
            text += sep + (('*%s*: ' % name) if (name and len(head) > 2) else '') + v
        out += wrap(text, '* ', '  ')
    return out


def reflow(lines):
    out, i, n = [], 0, len(lines)
    while i < n:
        ln = lines[i].rstrip('\n')
        if ln.lstrip().startswith('```'):
            out.append(ln)
            i += 1
            while i < n and not lines[i].lstrip().startswith('```'):
                out.append(lines[i].rstrip('\n'))
                i += 1
            if i < n:
                out.append(lines[i].rstrip('\n'))
                i += 1
            continue
        if ln.lstrip().startswith('|'):
            j = i
            while j < n and lines[j].lstrip().startswith('|'):
                j += 1
            rows = [l.rstrip('\n') for l in lines[i:j]]
            is_table = len(rows) >= 2 and re.match(r'^\s*\|?\s*:?-{2,}', rows[1].replace(' ', '')) is not None
            if is_table and not KEEP_TABLES and max(len(r) for r in rows) > WIDTH + 40:
                out += table_to_list(rows)
            else:
                out += rows
            i = j
            continue
        marker = re.match(r'^(\s*)((?:[*+-]|\d+\.)\s+|>\s*)?(.*)$', ln)
        indent, mark, body = marker.group(1), marker.group(2) or '', marker.group(3)
        if ln.startswith('#') or not ln.strip() or (ln.startswith('    ') and not mark):
            out.append(ln)
            i += 1
            continue
        # a block: this line and the lines that continue it (non-blank, no marker / heading / table / fence of their own)
        j = i + 1
        while j < n:
            nx = lines[j].rstrip('\n')
            if (not nx.strip() or nx.startswith('#') or nx.lstrip().startswith('|') or nx.lstrip().startswith('```')
                    or re.match(r'^\s*((?:[*+-]|\d+\.)\s+|>\s*)', nx) or nx.endswith('  ')
                    or (len(nx) - len(nx.lstrip()) != len(indent) + len(mark) and len(nx) - len(nx.lstrip()) != len(indent))):
                break
            j += 1
        block = [lines[k].rstrip('\n') for k in range(i, j)]
        if max(len(b) for b in block) <= WIDTH:
            out += block
        else:
            text = ' '.join([body] + [b.strip() for b in block[1:]])
            out += wrap(text, indent + mark, indent + ' ' * len(mark))
        i = j
    return out


for path in [a for a in sys.argv[1:] if not a.startswith('--')]:
    src = open(path).read().split('\n')
    res = reflow([l + '\n' for l in src])
    open(path, 'w').write('\n'.join(res).rstrip('\n') + '\n')
    over = sum(1 for l in res if len(l) > 200)
    print('%s: %d lines, %d still over 200 columns' % (path, len(res), over))
