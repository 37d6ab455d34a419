from .amplipy import main

main()
