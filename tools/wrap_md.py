#!/usr/bin/env python3
"""Re-wrap the prose of a markdown file to a column limit (default 120): paragraphs and list items are re-flowed with their
indentation, tables, headings, code fences, HTML and already-short lines are left alone.   usage: wrap_md.py FILE [WIDTH]"""
import re
import sys
import textwrap


def main():
    path = sys.argv[1]
    width = int(sys.argv[2]) if len(sys.argv) > 2 else 120
    out, para, in_code = [], [], False

    def flush():
        nonlocal para
        if not para:
            return
        first = para[0]
        m = re.match(r'^(\s*)((?:[-*+]|\d+\.)\s+)?', first)
        indent, bullet = m.group(1), m.group(2) or ''
        text = ' '.join([first[len(indent) + len(bullet):].strip()] + [l.strip() for l in para[1:]])
        sub = indent + ' ' * len(bullet)
        out.extend(textwrap.wrap(text, width=width, initial_indent=indent + bullet, subsequent_indent=sub,
                                 break_long_words=False, break_on_hyphens=False) or [''])
        para = []

    for line in open(path).read().split('\n'):
        s = line.rstrip()
        if s.lstrip().startswith('```'):
            flush()
            in_code = not in_code
            out.append(s)
            continue
        if in_code or not s.strip() or s.lstrip().startswith(('|', '#', '<', '>')) or re.match(r'^\s*[-=]{3,}\s*$', s):
            flush()
            out.append(s)
            continue
        if re.match(r'^\s*((?:[-*+]|\d+\.)\s+)', s):    # a new list item starts a new block
            flush()
        para.append(s)
    flush()
    open(path, 'w').write('\n'.join(out))


if __name__ == '__main__':
    main()
