from ._cli import main

main()
