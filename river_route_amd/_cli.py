"""`rr` command line (river_route/_cli.py:28-59): `rr route --router X cfg` or `rr X cfg` -> Router(cfg).route()."""
from __future__ import annotations

import argparse
import sys

from .routers import Muskingum, RapidMuskingum, UnitMuskingum

ROUTERS = {'Muskingum': Muskingum, 'RapidMuskingum': RapidMuskingum, 'UnitMuskingum': UnitMuskingum}

_HELP = {
    'Muskingum': 'Channel-only Muskingum routing (no lateral inflow)',
    'RapidMuskingum': 'RAPID-style Muskingum routing with lateral runoff',
    'UnitMuskingum': 'Unit hydrograph transform then Muskingum routing',
}


def main(argv=None) -> None:
    parser = argparse.ArgumentParser(prog='rr', description='river-route on MI355X: Muskingum river routing (HIP engine)')
    sub = parser.add_subparsers(dest='command')
    route = sub.add_parser('route', help='Run routing from a config file with a specified router')
    route.add_argument('config', type=str, help='Path to routing configuration file')
    route.add_argument('--router', type=str, required=True, choices=list(ROUTERS),
                       help='Router class to use (Muskingum, RapidMuskingum, or UnitMuskingum)')
    for name in ROUTERS:
        sub.add_parser(name, help=_HELP[name]).add_argument('config', type=str, help='Path to routing configuration file')
    args = parser.parse_args(argv)
    if args.command is None:
        parser.print_help()
        return
    router = ROUTERS.get(args.router if args.command == 'route' else args.command)
    if router is None:
        print(f'Unknown router: {args.router!r}. Must be one of: {", ".join(ROUTERS)}')
        sys.exit(1)
    router(args.config).route()


if __name__ == '__main__':
    main()
