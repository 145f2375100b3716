"""Re-wrap the long physical lines of a Markdown file at WIDTH columns without changing what it renders to:  python tools/wrap_md.py FILE [WIDTH=140]
Fenced code, tables and headings are left alone; a list item's continuation lines keep the item's text indent; every other long line becomes
several lines of the same paragraph.  Words are never split (citations like `file.py:12-34` stay whole)."""
import re
import sys
import textwrap


def wrap_file(path: str, width: int = 140) -> int:
    out, fenced, changed = [], False, 0
    for line in open(path).read().split("\n"):
        if line.lstrip().startswith("```"):
            fenced = not fenced
            out.append(line)
            continue
        if fenced or len(line) <= width or line.lstrip().startswith(("|", "#")) or line.startswith("    ") and not re.match(r"\s*([-*+]|\d+\.)\s", line):
            out.append(line)
            continue
        m = re.match(r"^(\s*)((?:[-*+]|\d+\.)\s+)?", line)
        indent, marker = m.group(1), m.group(2) or ""
        body = line[len(indent) + len(marker):]
        wrapped = textwrap.wrap(body, width=width - len(indent) - len(marker), break_long_words=False, break_on_hyphens=False)
        if not wrapped:
            out.append(line)
            continue
        out.append(indent + marker + wrapped[0])
        out.extend(indent + " " * len(marker) + w for w in wrapped[1:])
        changed += 1
    open(path, "w").write("\n".join(out))
    return changed


if __name__ == "__main__":
    print(wrap_file(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 140), "lines re-wrapped")
