import re, sys, textwrap
W = 158
def wrap(text, indent, first):
    return textwrap.wrap(' '.join(text.split()), width=W, initial_indent=first, subsequent_indent=indent, break_long_words=False, break_on_hyphens=False)
def table_items(buf):
    out = []
    for l in buf[2:]:
        cells = [c.strip() for c in re.split(r'(?<!\\)\|', l.strip().strip('|'))]
        body = ' — '.join(c for c in cells[1:] if c)
        out += wrap('**%s** — %s' % (cells[0], body), '  ', '* ')
    return out
def merge_paragraphs(md):
    """join the lines of plain paragraphs and of list items (continuation lines indented) so that wrapping starts from whole paragraphs"""
    out, in_code = [], False
    for l in md.split('\n'):
        special = (not l.strip()) or l.startswith(('|', '#', '>', '```')) or re.match(r'^\s*(?:[*-]|\d+\.)\s+', l)
        if l.startswith('```'):
            in_code = not in_code
        if in_code or special or not out or not out[-1].strip() or out[-1].startswith(('|', '#', '>', '```')):
            out.append(l)
        else:
            out[-1] = out[-1].rstrip() + ' ' + l.strip()
    return '\n'.join(out)
def reflow(md):
    md = merge_paragraphs(md)
    out, buf, in_code = [], [], False
    def flush():
        nonlocal buf
        if buf:
            if all(len(b) <= W for b in buf): out.extend(buf)
            else: out.extend(table_items(buf))
            buf = []
    for l in md.split('\n'):
        if l.startswith('```'):
            flush(); in_code = not in_code; out.append(l); continue
        if in_code:
            out.append(l); continue
        if l.startswith('|'):
            buf.append(l); continue
        flush()
        if len(l) <= W:
            out.append(l); continue
        m = re.match(r'^(\s*(?:[*-]|\d+\.)\s+)', l)
        if m:
            out += wrap(l[m.end():], ' ' * len(m.group(1)), m.group(1))
        elif l.startswith('#') or l.startswith('>'):
            out.append(l)
        else:
            ind = re.match(r'^\s*', l).group(0)
            out += wrap(l, ind, ind)
    flush()
    return '\n'.join(out)
if __name__ == '__main__':
    s = open(sys.argv[1]).read()
    t = reflow(s)
    open(sys.argv[1], 'w').write(t)
    ls = t.split('\n')
    print('lines', len(ls), 'longest', max(len(x) for x in ls), 'over160', sum(1 for x in ls if len(x) > 160))
